"""CPU: the SBA_WITH_OPENCV branch of the mirror class ("drops into main.cpp unchanged").

OpenCV is not in this image, so the branch cannot be linked or run here; what CAN be proven is that it parses and
type-checks: `g++ -fsyntax-only` against a minimal FAKE of the OpenCV declarations involved (tests/harness/fake_opencv,
declarations only).  Three levels:
  1. the mirror class + the OpenCV hooks header against a 3-line interface stand-in of the reference's matcher header;
  2. (when /root/reference is present) the same against the reference's OWN headers (spherical_surf.hpp ->
     feature_matcher.hpp -> debug_print.h), read in place;
  3. (when /root/reference is present) the reference's main/main.cpp, UNCHANGED and read in place through a symlink,
     with `../spherical_bundle_adjuster.hpp` resolving to the mirror header: the call sequence of main/main.cpp:29-32
     (constructor with 7 doubles, set_omp(omp_get_num_procs()), do_bundle_adjustment(Mat, Mat)) compiles as is.
"""
import subprocess
from pathlib import Path

import pytest

from helpers import ROOT

CSRC = ROOT / "spherical_bundle_adjuster_amd" / "csrc"
FAKE_CV = ROOT / "tests" / "harness" / "fake_opencv"
FAKE_REF = ROOT / "tests" / "harness" / "fake_reference"
REFERENCE = Path("/root/reference")


def syntax_only(src, *include_dirs, extra=()):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-fopenmp", "-Wall", "-Wextra", "-Wno-unknown-pragmas", "-DSBA_WITH_OPENCV"]
    for d in include_dirs:
        cmd += ["-I", str(d)]
    r = subprocess.run(cmd + list(extra) + [str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_opencv_branch_parses_against_interface_standin(tmp_path):
    syntax_only(CSRC / "spherical_bundle_adjuster.cpp", FAKE_CV, FAKE_REF)
    user = tmp_path / "hooks_user.cpp"
    user.write_text('#include "sba_opencv_hooks.hpp"\n'
                    "cv::Mat a(const cv::Mat& im) { return sba_cv::get_all(im, 600); }\n"
                    "cv::Mat b(const cv::Mat& im) { return sba_cv::crop_rotated_image(-90.f, im); }\n"
                    "void c(std::vector<cv::KeyPoint>& k) { sba_cv::rotate_keypoint(45.f, k, 3840, 1920); "
                    "sba_cv::cube2equi_keypoints(k, 600, 3840, 1920); }\n")
    syntax_only(user, FAKE_CV, CSRC)
    # the default-matcher branch is really there: without the macro the same file must NOT mention spherical_surf
    txt = (CSRC / "spherical_bundle_adjuster.cpp").read_text()
    assert "spherical_surf fm;" in txt and "fm.set_omp(this->num_proc);" in txt and "#ifdef SBA_WITH_OPENCV" in txt


@pytest.mark.skipif(not (REFERENCE / "main" / "main.cpp").exists(), reason="reference tree not present")
def test_reference_main_cpp_compiles_unchanged_against_the_mirror(tmp_path):
    # level 2: the mirror against the reference's own matcher headers
    syntax_only(CSRC / "spherical_bundle_adjuster.cpp", FAKE_CV, REFERENCE)
    # level 3: main/main.cpp as it lies in the reference; its `#include "../spherical_bundle_adjuster.hpp"` must find the
    # mirror header, so both are symlinked into a scratch tree of the same shape (nothing is copied)
    (tmp_path / "main").mkdir()
    (tmp_path / "main" / "main.cpp").symlink_to(REFERENCE / "main" / "main.cpp")
    (tmp_path / "spherical_bundle_adjuster.hpp").write_text(
        f'#include "{CSRC / "spherical_bundle_adjuster.hpp"}"\n')     # forwarding header, like an installed copy
    syntax_only(tmp_path / "main" / "main.cpp", FAKE_CV, REFERENCE)
