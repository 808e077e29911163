"""MI355X-native spherical bundle-adjustment hot path (HIP/CDNA4) behind the reference's
``spherical_bundle_adjuster`` interface.  See DESIGN.md / INTEGRATION.md.

Importing the package does not load the HIP library; the first API call does and fails loudly
(``LibraryNotBuilt``) if ``libsba_hip.so`` has not been built."""
from ._cabi import ABI_VERSION, LibraryNotBuilt, SbaError, load_library  # noqa: F401

__all__ = ["ABI_VERSION", "LibraryNotBuilt", "SbaError", "load_library"]
