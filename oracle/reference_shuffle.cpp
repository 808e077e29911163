// ORACLE (test infrastructure; only tests/, smoke() and bench.py's cpu_baseline may use anything under oracle/).
//
// The reference's random_array (spherical_bundle_adjuster.hpp:182-211) restated with the VERY library call the reference
// makes: std::iota + std::random_shuffle(first, last) -- libstdc++'s two-iterator overload, which draws from the
// process-wide std::rand() -- one fresh permutation per trial, the first sample_n = int(n * 0.25) entries used
// (initial_guess, spherical_bundle_adjuster.cpp:130-141).  This is the one place where the oracle is pinned by the
// reference's own runtime library rather than by a restatement of it: the product's hand-written loop
// (csrc/sba_epipolar.hpp: reference_trial_subsets) must reproduce these lists element for element from the same rand()
// state.  Own translation unit: std::random_shuffle is deprecated since C++14 and gone from C++17's <algorithm> contract,
// so this file is built as the reference is -- the compiler's default dialect of its day (gnu++14), no warnings wanted.
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <vector>

extern "C" {
// reseed = 1: srand(1) first -- the state of a process that never called srand() (C standard, 7.22.2.2) and has drawn nothing yet.
void orc_reference_trial_subsets(int n, int trials, int reseed, int* out, int* sample_n_out) {
  if (reseed) std::srand(1);
  const int sample_n = n * 0.25;                     // `int sample_n = match_size*0.25;` (.cpp:133)
  if (sample_n_out) *sample_n_out = sample_n;
  if (!out || n <= 0) return;
  std::vector<int> perm(n);
  for (int t = 0; t < trials; ++t) {
    // what constructing a random_array(n) does (.hpp:206-210), then sample_n calls of get_rand() (.hpp:194-200: entries
    // 0, 1, 2, ... of the permutation, sample_n <= n so the counter never wraps)
    std::iota(perm.begin(), perm.end(), 0);
    std::random_shuffle(perm.begin(), perm.end());
    for (int i = 0; i < sample_n; ++i) out[static_cast<long>(t) * sample_n + i] = perm[i];
  }
}
void orc_srand(unsigned seed) { std::srand(seed); }
int orc_rand(void) { return std::rand(); }
}
