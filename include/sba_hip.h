/*
 * sba_hip.h -- C-ABI drop-in boundary of the MI355X spherical bundle-adjustment hot path.
 *
 * What this replaces in the reference (whdlgp/spherical_bundle_adjuster, all citations
 * relative to the reference tree):
 *
 *   - the four `add_residual` loops that build one Ceres residual block per match
 *     (spherical_bundle_adjuster.cpp:870-889, :921-945, :978-1002, :1034-1063)
 *       -> sba_problem_upload()            (one flat upload instead of 3 heap objects per match)
 *   - one residual+Jacobian sweep of ceres::Solve over those blocks, i.e. the templated
 *     functors (spherical_bundle_adjuster.cpp:843-868, :891-919, :947-976) pushed through
 *     AutoDiffCostFunction + HuberLoss(1.0) and accumulated into J^T J / J^T r
 *       -> sba_problem_eval() / sba_problem_eval_pack()
 *   - `ceres::Solve(opt, &problem_rot|tran, &summary)` inside solve_problem()
 *     (spherical_bundle_adjuster.cpp:183-217, options :334-338)
 *       -> sba_problem_solve()             (Levenberg-Marquardt on the host, normal equations from the GPU)
 *   - the d-only stage (spherical_bundle_adjuster.cpp:1004-1063)
 *       -> sba_problem_solve_depths()
 *   - pixel -> unit sphere (spherical_bundle_adjuster.cpp:271-298)
 *       -> sba_keypoints_to_sphere()
 *   - equi2cube::get_all (equi2cube.cpp:12-302)
 *       -> sba_equi2cube(), sba_equi2cube_device()
 *   - initial_guess / eight_point_estimation (spherical_bundle_adjuster.cpp:47-181)
 *       -> sba_problem_initial_guess()
 *   - spherical_surf::rotate_keypoint / crop_rotated_image, equi2cube_surf::cube2equi_pixel
 *       -> sba_rotate_keypoints(), sba_crop_rotated_image(), sba_cube2equi_keypoints()
 *   - one run of main/main.cpp per image pair
 *       -> sba_batch_*()                   (many pairs per launch, one LM per pair)
 *
 * Conventions
 *   - every function returns an int status: 0 = SBA_OK, negative = error; the text of the
 *     last error on the calling thread is returned by sba_last_error().
 *   - no exceptions cross this boundary, no C++ or torch types appear in a signature.
 *   - the caller owns every host array it passes in; the library owns the device buffers
 *     inside the opaque handles.  Handles are not thread-safe.
 *   - residual convention (spherical_bundle_adjuster.cpp:897-916):
 *         e_i = d2 * x2_i - ( R(rot) * (d1 * x1_i) - tran )
 *     with R(rot) the angle-axis rotation (Ceres AngleAxisRotatePoint semantics, including
 *     its small-angle branch), robustified per 3-vector block by Huber(delta).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with SBA_ERR_NO_DEVICE.
 */
#ifndef SBA_HIP_H_
#define SBA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBA_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------------------ */
enum {
  SBA_OK = 0,
  SBA_ERR_INVALID_ARG = -1,
  SBA_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime unusable                     */
  SBA_ERR_HIP = -3,         /* a HIP call failed (text in sba_last_error)               */
  SBA_ERR_NOT_UPLOADED = -4,
  SBA_ERR_COMM = -5,        /* RCCL / user all-reduce hook failed                        */
  SBA_ERR_NUMERIC = -6,     /* non-finite normal equations, singular system             */
  SBA_ERR_UNSUPPORTED = -7
};

/* ---- which parameter block is free (= which reference functor is evaluated) ---------- */
enum {
  SBA_MODE_ROT = 0,  /* ba_spherical_costfunctor_rot_only  (.cpp:891-919): rot free, tran frozen */
  SBA_MODE_TRAN = 1, /* ba_spherical_costfunctor_tran_only (.cpp:947-976): tran free, rot frozen */
  SBA_MODE_RT = 2    /* ba_spherical_costfunctor (.cpp:843-868) with d held fixed: rot+tran free */
};

/* ---- where d1,d2 come from ----------------------------------------------------------- */
enum {
  SBA_DEPTH_UNIFORM = 0,  /* two scalars for every match: what the reference actually does,
                             init_d[0][0] / init_d[1][0] (.cpp:941-942, :998-999)          */
  SBA_DEPTH_PER_MATCH = 1 /* d12[i][0], d12[i][1] per match (.cpp:887, the joint functor)  */
};

/* ---- storage type of the device-resident unit vectors -------------------------------- */
enum {
  SBA_STORE_F64 = 0, /* 48 B / correspondence (64 B with per-match depths): reference-faithful */
  SBA_STORE_F32 = 1  /* 24 B / correspondence (32 B): arithmetic stays f64                    */
};

/* ---- which sweep kernel evaluates the Jacobian terms (results agree to rounding) ---------------- */
enum {
  SBA_KERNEL_FACTORED = 0, /* default: d e/d rot = -[v]x J_l(rot), v = -d1 R x1; the device accumulates the
                              moments sum w v v^T, sum w v e^T, sum w v and the host applies J_l          */
  SBA_KERNEL_EXPLICIT = 1  /* the 3x3 Jacobian block d e/d rot is formed per match on the device          */
};

/* ---- translation parameterisation in SBA_MODE_RT / SBA_MODE_TRAN ---------------------- */
enum {
  SBA_TRAN_FREE = 0,   /* 3 free components, additive update (reference behaviour)            */
  SBA_TRAN_SPHERE = 1  /* 5-DoF: tran stays on the sphere |tran| = const, 2-dim tangent update */
};

/* Indices into the 24-double device pack (the thing that is all-reduced across GPUs).     */
enum {
  SBA_PACK_HAA = 0,   /* [0..5]  upper triangle of sum w A^T A : 00 01 02 11 12 22          */
  SBA_PACK_HAT = 6,   /* [6..14] sum w A^T  (3x3 row-major: row = rot index, col = tran idx) */
  SBA_PACK_SW = 15,   /* sum w            (H_tt = SW * I3)                                  */
  SBA_PACK_GA = 16,   /* [16..18] sum w A^T e                                               */
  SBA_PACK_GT = 19,   /* [19..21] sum w e                                                   */
  SBA_PACK_COST = 22, /* 1/2 sum rho(|e|^2)                                                 */
  SBA_PACK_NOUT = 23, /* number of blocks in Huber's outlier region (as a double)           */
  SBA_PACK_SIZE = 24
};

/* Expanded normal equations over the parameter order [rot0 rot1 rot2 tran0 tran1 tran2].  */
typedef struct sba_normal_eq {
  double H[36];      /* row-major symmetric 6x6 = sum rho' J^T J; unused blocks are zero     */
  double g[6];       /* sum rho' J^T e                                                       */
  double cost;       /* 1/2 sum rho(|e|^2)   (what Ceres reports as the cost)                */
  double sum_w;      /* sum rho'                                                             */
  double n_outlier;  /* blocks with |e|^2 > delta^2                                          */
} sba_normal_eq;

/* Levenberg-Marquardt options; defaults (sba_lm_options_default) restate the Ceres defaults
 * that govern the reference's solve (spherical_bundle_adjuster.cpp:334-338 sets only
 * max_num_iterations = 50, the linear solver, stdout logging and the thread count).       */
typedef struct sba_lm_options {
  int max_num_iterations;             /* 50  (.cpp:336)                                     */
  double initial_trust_region_radius; /* 1e4                                               */
  double max_trust_region_radius;     /* 1e16                                              */
  double min_trust_region_radius;     /* 1e-32                                             */
  double min_relative_decrease;       /* 1e-3                                              */
  double min_lm_diagonal;             /* 1e-6                                              */
  double max_lm_diagonal;             /* 1e32                                              */
  double function_tolerance;          /* 1e-6                                              */
  double gradient_tolerance;          /* 1e-10                                             */
  double parameter_tolerance;         /* 1e-8                                              */
  int jacobi_scaling;                 /* 1                                                 */
  double huber_delta;                 /* 1.0 (.cpp:887,:943,:1000); <= 0 disables the loss */
  int tran_param;                     /* SBA_TRAN_FREE (reference) or SBA_TRAN_SPHERE       */
  int verbose;                        /* 1 = per-iteration line on stdout (.cpp:337)        */
  /* Projected line search of a bounds-constrained problem -- only the d-only stage has bounds
   * (.cpp:1060-1061).  Ceres runs it on every trust-region step when the count below is > 0; the
   * reference leaves all five at Ceres' defaults.  ARMIJO, CUBIC interpolation.  0 = off.        */
  int max_num_line_search_step_size_iterations;    /* 20                                     */
  double line_search_sufficient_function_decrease; /* 1e-4                                   */
  double max_line_search_step_contraction;         /* 1e-3                                   */
  double min_line_search_step_contraction;         /* 0.6                                    */
  double min_line_search_step_size;                /* 1e-9                                   */
} sba_lm_options;

enum {
  SBA_TERM_CONVERGENCE_FUNCTION = 1,
  SBA_TERM_CONVERGENCE_GRADIENT = 2,
  SBA_TERM_CONVERGENCE_PARAMETER = 3,
  SBA_TERM_NO_CONVERGENCE = 4,     /* iteration limit                                       */
  SBA_TERM_MIN_RADIUS = 5,
  SBA_TERM_FAILURE = 6
};

typedef struct sba_lm_summary {
  int termination;            /* SBA_TERM_*                                                */
  int num_iterations;         /* LM iterations run (successful + unsuccessful)             */
  int num_successful_steps;
  int num_evaluations;        /* residual+Jacobian sweeps over the correspondences         */
  double initial_cost;
  double final_cost;
  double final_gradient_max_norm;
  double final_radius;
  double seconds_total;       /* wall clock of the whole solve                             */
  double seconds_eval;        /* of which: waiting for the device sweeps                   */
  int num_line_search_steps;  /* d-only stage: step-size contractions of the projected line search */
} sba_lm_summary;

typedef struct sba_problem sba_problem; /* opaque: one shard of correspondences on one GPU  */

/* User all-reduce hook (sum, in place) over `count` doubles at device address `device_buf`,
 * to be enqueued on HIP stream `stream`.  Lets a host that already owns a communicator
 * (e.g. torch.distributed) supply the exchange; see also sba_problem_comm_init_rank.        */
typedef int (*sba_allreduce_fn)(void* device_buf, size_t count, void* stream, void* user);

/* ---- library ------------------------------------------------------------------------- */
int sba_abi_version(void);
const char* sba_last_error(void);
int sba_device_count(int* count);
void sba_lm_options_default(sba_lm_options* opt);

/* ---- problem life cycle ---------------------------------------------------------------- */
/* device: HIP ordinal.  stream: an existing hipStream_t to launch on, or NULL to let the
 * problem create its own non-blocking stream.                                              */
int sba_problem_create(sba_problem** out, int device, void* stream);
int sba_problem_destroy(sba_problem* p);

/* Upload n correspondences.  left_xyz / right_xyz: packed double[3n], exactly
 * `std::vector<cv::Point3d>::data()` of key_point_left_rect / key_point_right_rect
 * (spherical_bundle_adjuster.cpp:286-298).  d12: NULL, or double[2n] =
 * `std::vector<std::array<double,2>>::data()` (init_d, .cpp:311-327) for per-match depths.
 * The arrays are copied (and re-laid-out as planes) once; they are not referenced afterwards.
 * n == 0 is legal (every eval then returns zeros).                                          */
int sba_problem_upload(sba_problem* p, const double* left_xyz, const double* right_xyz,
                       const double* d12, size_t n, int store);

/* Same, from arrays already resident on this problem's device (same AoS layouts).          */
int sba_problem_upload_device(sba_problem* p, const void* left_xyz_dev, const void* right_xyz_dev,
                              const void* d12_dev, size_t n, int store);

/* Same, straight from the matcher's output: `left_keypoints[i]` / `right_keypoints[i]` are the MATCHED
 * key-point records (cv::KeyPoint layout: the first two float32 of each record are pt.x, pt.y;
 * stride_bytes = sizeof(cv::KeyPoint) = 28) of the two equirectangular images.  The pixel -> unit-sphere
 * map of spherical_bundle_adjuster.cpp:271-298 runs on the device and writes the coordinate planes
 * directly, so the host never materialises the cv::Point3d arrays.       */
int sba_problem_upload_keypoints(sba_problem* p, const void* left_keypoints, const void* right_keypoints, size_t n,
                                 size_t stride_bytes, int im_width, int im_height, const double* d12, int store);

int sba_problem_size(const sba_problem* p, size_t* n);
/* (Re)upload only the per-match depths d12 (double[2n], init_d layout) of the resident correspondences.    */
int sba_problem_set_depths(sba_problem* p, const double* d12);
/* Select the sweep kernel (SBA_KERNEL_*); also settable with the environment variable
 * SBA_KERNEL=explicit at sba_problem_create time.                                                 */
int sba_problem_set_kernel(sba_problem* p, int kind);

/* ---- one residual + Jacobian sweep ------------------------------------------------------ */
/* Evaluates all local correspondences at (rot, tran), reduces on the device, all-reduces if a
 * communicator/hook is installed, and returns the expanded normal equations.  depth_mode
 * SBA_DEPTH_UNIFORM uses (d1, d2) for every match; SBA_DEPTH_PER_MATCH uses the uploaded d12
 * (d1, d2 ignored).  huber_delta <= 0 means no robustifier.  Synchronous.                  */
int sba_problem_eval(sba_problem* p, int mode, int depth_mode, const double rot[3],
                     const double tran[3], double d1, double d2, double huber_delta,
                     sba_normal_eq* out);

/* As above but returns the raw 24-double pack (SBA_PACK_* layout).                         */
int sba_problem_eval_pack(sba_problem* p, int mode, int depth_mode, const double rot[3],
                          const double tran[3], double d1, double d2, double huber_delta,
                          double pack[SBA_PACK_SIZE]);

/* Enqueue `repeat` back-to-back sweeps without host synchronisation in between and time them with
 * HIP events recorded on the problem's stream: *mean_step_ms = (sweep + finalize [+ all-reduce] +
 * publication) per repeat; *mean_sweep_ms = the sweep kernel alone, `repeat` launches back to back under
 * one event pair (kernel + the boundary between dependent launches).  Either output pointer may be NULL.
 * The last sweep's pack is returned.  This is bench.py's timing primitive.                          */
int sba_problem_eval_timed(sba_problem* p, int mode, int depth_mode, const double rot[3],
                           const double tran[3], double d1, double d2, double huber_delta,
                           int repeat, double pack[SBA_PACK_SIZE], double* mean_step_ms,
                           double* mean_sweep_ms);

/* The sweep kernel launched `repeat` times back to back with a HIP event recorded between every two launches on the
 * problem's stream: launch_ms[i] = device time from the event before launch i to the event after it (the kernel plus
 * one event boundary).  Shows the spread of the dominant kernel -- first launches after idle run at ramping clocks --
 * that the mean of sba_problem_eval_timed hides.  repeat <= 4096.                                            */
int sba_problem_eval_launch_times(sba_problem* p, int mode, int depth_mode, const double rot[3],
                                  const double tran[3], double d1, double d2, double huber_delta,
                                  int repeat, float* launch_ms);

/* `steps` complete, host-synchronous sweeps in a row -- each one exactly what an LM iteration costs (launch,
 * reduction, all-reduce if installed, result on the host before the next launch) -- without a language binding
 * between them.  Returns the last pack and the wall-clock seconds of the loop.                               */
int sba_problem_eval_steps(sba_problem* p, int mode, int depth_mode, const double rot[3],
                           const double tran[3], double d1, double d2, double huber_delta, int steps,
                           double pack[SBA_PACK_SIZE], double* seconds);

/* Host-only: expand a pack into the 6x6 system (no device needed).                         */
int sba_expand_pack(int mode, const double pack[SBA_PACK_SIZE], sba_normal_eq* out);

/* ---- Levenberg-Marquardt solve (replaces ceres::Solve for the rot / tran / joint stage) -- */
/* rot and tran are updated in place like init_rot / init_tran in the reference
 * (spherical_bundle_adjuster.cpp:202-209).                                                  */
int sba_problem_solve(sba_problem* p, int mode, int depth_mode, double rot[3], double tran[3],
                      double d1, double d2, const sba_lm_options* opt, sba_lm_summary* summary);

/* d-only stage (spherical_bundle_adjuster.cpp:1004-1063, `Solve(opt, &problem_d, &summary)` at :197): ONE
 * bounded trust-region problem over all per-match depth pairs (5 residuals per match: the reprojection
 * 3-vector and lambda*exp(-c*d1), lambda*exp(-c*d2); no loss; lower bound 0; the reference uses lambda = c = 1).
 * Needs per-match depths uploaded (they are the initial values, init_d) and updates them on the device, so a
 * following SBA_DEPTH_PER_MATCH sweep sees the refined depths.  d12_out (double[2n], may be NULL) receives
 * them in init_d layout.  opt NULL = defaults (huber_delta / tran_param are ignored).  Every trust-region step
 * goes through Ceres' projected Armijo line search first (the problem is bounds-constrained and the reference
 * leaves max_num_line_search_step_size_iterations at 20): the first trial, step size 1, is evaluated by the same
 * device pass that computes the step; each contraction costs one more pass.  With a transport attached (sharded
 * problem) the nine global reductions of every pass are all-reduced, all ranks take the same steps, and every
 * rank receives its own shard's depths (at most 8 shards, see sba_problem_set_shard).                       */
int sba_problem_solve_depths(sba_problem* p, const double rot[3], const double tran[3], double lambda,
                             double c, const sba_lm_options* opt, double* d12_out, sba_lm_summary* summary);

/* ---- multi-GPU: one process (and one sba_problem) per GPU, correspondences sharded ------ */
/* Option A: native RCCL.  Rank 0 calls sba_comm_unique_id, ships the 128 bytes to the other
 * ranks by any host channel, then every rank calls sba_problem_comm_init_rank.  After that
 * every eval / solve all-reduces the 24-double pack once (ncclAllReduce, ncclDouble, ncclSum). */
#define SBA_COMM_ID_BYTES 128
int sba_comm_unique_id(char id[SBA_COMM_ID_BYTES]);
int sba_problem_comm_init_rank(sba_problem* p, int nranks, int rank, const char id[SBA_COMM_ID_BYTES]);
/* 1 if librccl could be loaded and bound in this process (0: reason in sba_last_error).  ncclCommInitRank is itself
 * a collective: a rank that cannot even load the library would leave the others waiting inside it, so ranks agree on
 * this BEFORE any of them calls sba_problem_comm_init_rank.                                                        */
int sba_rccl_available(void);
/* Tear the communicator down again (after a rank reported failure: all ranks destroy theirs and fall back together). */
int sba_problem_comm_destroy(sba_problem* p);
/* Option C: direct peer exchange, no collective library on the data path.  Every rank owns a 4 KiB inbox in
 * fine-grained device memory that all peers map through HIP IPC; per sweep one wave stores its 24 doubles into every
 * peer's inbox (over xGMI), polls its own inbox and sums the nranks contributions in rank order (bit-identical on all
 * ranks), then publishes to the host.  Set-up: every rank calls sba_problem_peer_export (allocates the inbox, returns
 * its 64-byte IPC handle), the handles are all-gathered by any host channel into handles[nranks][64], every rank calls
 * sba_problem_peer_connect, then all ranks together sba_problem_peer_selftest; on any failure call
 * sba_problem_peer_disable everywhere and use option A.  At most 8 ranks (one node).                            */
#define SBA_PEER_HANDLE_BYTES 64
int sba_problem_peer_export(sba_problem* p, int nranks, int rank, char handle[SBA_PEER_HANDLE_BYTES]);
int sba_problem_peer_connect(sba_problem* p, const char* handles);
int sba_problem_peer_selftest(sba_problem* p, int rounds, int* ok);
int sba_problem_peer_disable(sba_problem* p);
/* Option B: user hook (e.g. torch.distributed.all_reduce on a tensor aliasing device_buf).  */
int sba_problem_set_allreduce(sba_problem* p, sba_allreduce_fn fn, void* user);
/* Which shard of the correspondences this problem holds.  sba_problem_comm_init_rank and sba_problem_peer_export set
 * it themselves; with the user hook call it explicitly before sba_problem_solve_depths, whose two max-norms
 * (projected gradient, step) travel through the SUM all-reduce as one pack slot per shard each (hence at most 8
 * shards).                                                                                                         */
int sba_problem_set_shard(sba_problem* p, int rank, int nranks);
/* Device address of the 24-double result pack the hook / RCCL operates on (a sum over
 * correspondences in either kernel's layout, so summing it across shards is exact).            */
int sba_problem_pack_device_ptr(sba_problem* p, void** dev_ptr);

/* Host threads for the host-side loops that have any (the trials of the initial guess; the per-pair LM steps of
 * sba_batch_solve, from 64 pairs per thread) -- the counterpart of
 * the reference's set_omp(num_proc) (.cpp:835-841).  0 = one per hardware thread (at most 16); default 1, or
 * SBA_HOST_THREADS.  Process-wide; results do not depend on it.  A trial costs a few microseconds, so threads are
 * only started when there are at least 256 trials per thread (the reference's 80 trials run serially).           */
int sba_set_host_threads(int n);

/* ---- 8-point initial guess (eight_point_estimation / initial_guess, spherical_bundle_adjuster.cpp:47-181) ---- */
/* Device part: one pass over the uploaded correspondences accumulating A^T A of the rows kron(left_i, right_i)
 * (.cpp:53-68) for 64 interleaved groups, group(i) = (i / 2) % 64.  groups: double[64][45] (upper triangle,
 * row-major a <= b).  With a transport attached (sharded problem) the sums are all-reduced: every rank receives
 * the moments of the whole problem (group g = the union of all shards' group g) and so derives the same guess.  */
int sba_problem_epipolar_moments(sba_problem* p, double* groups);
/* Host part (no device needed): `trials` trials (80 in the reference, .cpp:130), each on a random
 * `subset_fraction` (0.25, .cpp:133) of the groups: null vector of A (smallest eigenvector of A^T A), rank-2
 * projection, decomposeEssentialMat, Euler angles, validity (< 1.57), consensus pick by 20-80 % trimmed mean
 * distance.  Outputs R_vec_out (Euler angles) and T_vec_out of the reference; the reference then starts the BA
 * from init_rot = -R_vec_out, init_tran = T_vec_out (.cpp:330-331).                                          */
int sba_initial_guess_from_moments(const double* groups, int trials, double subset_fraction,
                                   unsigned long long seed, double rot_euler[3], double tran[3],
                                   int* num_candidates);
/* Both parts.  SBA_GUESS_SAMPLING=reference in the environment routes this call to sba_problem_initial_guess_reference
 * (seed is then unused) for problems it supports.                                                            */
int sba_problem_initial_guess(sba_problem* p, int trials, double subset_fraction, unsigned long long seed,
                              double rot_euler[3], double tran[3], int* num_candidates);

/* ---- the same initial guess from the REFERENCE'S OWN random subsets (small problems) ------------------------- */
/* The random stream behind it.  The reference shuffles with std::random_shuffle on rand(), which it never seeds: glibc's
 * generator in its srand(1) state.  The library keeps its OWN copy of that generator (same algorithm, same values --
 * pinned against the real rand() by the tests), starting in the never-seeded state and running on from call to call like
 * the reference's stream does from image pair to image pair.  It does not borrow the process's rand(): once HIP is
 * initialised that stream is no longer the caller's alone (libhsa-runtime64 imports srand / rand).  SBA_GUESS_RAND=libc in
 * the environment switches to the process's rand() -- then draws made by other code of the process (a FLANN matcher's
 * kd-trees, as in the reference's own process) count, and so do the ROCm runtime's.
 * sba_reference_rand_seed(1) puts the stream back into the never-seeded state (any seed: as srand(seed));
 * sba_reference_rand_next() draws one value, as rand() would.                                                    */
int sba_reference_rand_seed(unsigned int seed);
int sba_reference_rand_next(void);
/* Host-only (no device needed): the match indices the reference's `trials` trials draw for a problem of n matches --
 * random_array (spherical_bundle_adjuster.hpp:182-211: std::iota + std::random_shuffle) constructed once per trial, its
 * first sample_n = (int)(n * subset_fraction) entries used (spherical_bundle_adjuster.cpp:130-141).  indices:
 * int[trials][sample_n].  Draws (n - 1) values per trial from the stream above, in libstdc++'s order.  *sample_n_out
 * receives sample_n (pass indices = NULL to query it without drawing).                                          */
int sba_reference_trial_subsets(int n, int trials, double subset_fraction, int* indices, int* sample_n_out);
/* Device part: A^T A of the rows kron(left_i, right_i) over each trial's index list.  indices: int[trials][sample_n]
 * (host), every entry < n; moments: double[trials][45] (host; upper triangle, row-major a <= b).  Single-GPU problems
 * only (an index list addresses the resident shard).                                                           */
int sba_problem_epipolar_subset_moments(sba_problem* p, const int* indices, int trials, int sample_n, double* moments);
/* Both parts + the consensus: what initial_guess (.cpp:118-181) computes, from the subsets the reference itself would
 * draw.  At most SBA_REFERENCE_SAMPLING_MAX_N matches (the reference's problem sizes; larger problems use the group
 * sampling of sba_problem_initial_guess), at least 4 (sample_n >= 1).                                          */
#define SBA_REFERENCE_SAMPLING_MAX_N (1 << 20)
int sba_problem_initial_guess_reference(sba_problem* p, int trials, double subset_fraction, double rot_euler[3],
                                        double tran[3], int* num_candidates);

/* ---- batched per-pair solve (BASELINE config C5: many ERP pairs, each its own two-view problem) ------- */
/* The reference handles one image pair per process run (main/main.cpp:6-34); a batch holds `num_pairs`
 * independent problems on one GPU: pair g owns correspondences [offsets[g], offsets[g+1]) of the
 * concatenated arrays (same layouts as sba_problem_upload; offsets has num_pairs+1 entries, ragged and empty
 * pairs allowed).  ONE sweep launch evaluates every pair at its own (rot, tran) -- the per-pair sweep state and the
 * mapping of the reduced moments to normal equations are computed on the device as well; sba_batch_solve runs one LM per
 * pair in lock-step (same schedule as sba_problem_solve).  Pairs are independent, so across GPUs they are
 * simply split between processes -- no collective.                                                        */
typedef struct sba_batch sba_batch;
int sba_batch_create(sba_batch** out, int device, void* stream);
int sba_batch_destroy(sba_batch* b);
int sba_batch_set_kernel(sba_batch* b, int kind);
int sba_batch_upload(sba_batch* b, const double* left_xyz, const double* right_xyz, const double* d12,
                     const size_t* offsets, int num_pairs, int store);
/* Re-send only the per-match depths (same layout as the d12 of sba_batch_upload, which must have carried depths): the
 * counterpart of sba_problem_set_depths -- sba_batch_solve_depths / sba_batch_solve_problem refine the depths in place, the
 * coordinates stay resident.  (sba_batch_upload with unchanged offsets / store keeps every allocation too and only moves data.) */
int sba_batch_set_depths(sba_batch* b, const double* d12);
int sba_batch_size(const sba_batch* b, int* num_pairs, int* blocks_per_pair);
/* rot, tran: double[num_pairs][3]; d1, d2: double[num_pairs] uniform depths per pair (NULL = 1.0; ignored with
 * SBA_DEPTH_PER_MATCH); packs: double[num_pairs][SBA_PACK_SIZE].                                          */
int sba_batch_eval(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                   const double* d1, const double* d2, double huber_delta, double* packs);
/* `steps` host-synchronous batched steps in a C loop (bench / tuning): wall-clock mean per step and its split into
 * host preparation of the per-pair R|t state, launch-to-result on the device, and host conversion of the packs.  */
int sba_batch_eval_timed(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                         const double* d1, const double* d2, double huber_delta, int steps, double* packs,
                         double* mean_step_ms, double* mean_prepare_ms, double* mean_device_ms,
                         double* mean_convert_ms);
/* The batched sweep kernel alone, `repeat` launches with a HIP event between every two (as
 * sba_problem_eval_launch_times): launch_ms[i] = device time of launch i.  repeat <= 4096.                        */
int sba_batch_sweep_launch_times(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                                 const double* d1, const double* d2, double huber_delta, int repeat, float* launch_ms);
/* The same measurement on the dominant kernel of the step this batch really runs: with one block per pair the whole
 * step is ONE launch (batch_step_kernel: per-pair state read from mapped host memory, sweep, fold, conversion, packs and
 * sequence word published to the host -- launched here exactly as a step launches it, only the host does not wait
 * between launches), otherwise the batched sweep kernel as above.  sba_batch_step_is_fused: 1 / 0 (negative =
 * error) -- which of the two it is (SBA_BATCH_FUSED_STEP=0 in the environment keeps the three-kernel chain).        */
int sba_batch_step_launch_times(sba_batch* b, int mode, int depth_mode, const double* rot, const double* tran,
                                const double* d1, const double* d2, double huber_delta, int repeat, float* launch_ms);
int sba_batch_step_is_fused(const sba_batch* b);
/* One Levenberg-Marquardt solve per pair (what sba_problem_solve does for one problem).  rot / tran are updated in place per
 * pair; summaries (sba_lm_summary[num_pairs]) and status (int[num_pairs], SBA_OK or SBA_ERR_NUMERIC per pair) may be NULL.
 * With one block per pair the solvers run on the device: one launch covers the first sweeps of every pair, pairs that need
 * more go on in per-iteration launches whose blocks are dealt out to the pairs still iterating (DESIGN.md section 3.5;
 * SBA_BATCH_DYNAMIC / SBA_BATCH_LM_FIRST_SWEEPS in INTEGRATION.md section 7).                                          */
int sba_batch_solve(sba_batch* b, int mode, int depth_mode, double* rot, double* tran, const double* d1,
                    const double* d2, const sba_lm_options* opt, sba_lm_summary* summaries, int* status);

/* The 8-point initial guess (initial_guess, spherical_bundle_adjuster.cpp:47-181) of every pair of the batch -- what running
 * the reference once per pair does first.  Device part: the 64 x 45 group moments of every pair in one launch (one block per
 * pair; group = (match index WITHIN the pair / 2) % 64, as sba_problem_epipolar_moments on that pair alone; the sums differ
 * from it in summation order only).  groups: double[num_pairs][64][45].                                              */
int sba_batch_epipolar_moments(sba_batch* b, double* groups);
/* Both parts: pair g's result is what sba_initial_guess_from_moments makes of pair g's moments with the same trials /
 * subset_fraction / seed.  rot_euler, tran: double[num_pairs][3] (R_vec_out, T_vec_out of the reference, per pair);
 * num_candidates (may be NULL): int[num_pairs]; status (may be NULL): int[num_pairs], SBA_OK or SBA_ERR_NUMERIC (no valid
 * rotation candidate: that pair's outputs are zeros).  Returns SBA_ERR_NUMERIC when any pair failed.                  */
int sba_batch_initial_guess(sba_batch* b, int trials, double subset_fraction, unsigned long long seed, double* rot_euler,
                            double* tran, int* num_candidates, int* status);

/* The reference's whole per-pair pipeline -- do_bundle_adjustment's initial values and solve_problem
 * (spherical_bundle_adjuster.cpp:302-331, :183-217) -- for EVERY pair of the batch: what running main/main.cpp once per ERP
 * pair computes after matching.  use_initial_guess != 0: the 8-point consensus guess (trials, subset_fraction, seed as
 * sba_batch_initial_guess), init_rot = -R_vec_out, init_tran = T_vec_out (.cpp:330-331); a pair without a valid candidate
 * keeps the caller's rot / tran and is reported in status.  use_initial_guess == 0: rot / tran are the start values (the
 * alternative the reference keeps in a comment, .cpp:328-329).  Then the d-only stage on the uploaded depths (lambda = c = 1,
 * .cpp:1057-1058), the rot-only and the tran-only stage with init_d[0][0], init_d[1][0] of the pair as the depths of every
 * match (.cpp:941-942, :998-999).  rot, tran: double[num_pairs][3], in (start values) / out (result).  All of the following
 * may be NULL: d12_out (refined depths, as sba_batch_solve_depths), d_uniform double[num_pairs][2] (the two depths the
 * last stages used), guess_candidates int[num_pairs], the three summary arrays [num_pairs], status int[num_pairs] (first
 * failing stage's code per pair).  Returns SBA_ERR_NUMERIC when any pair failed; the other pairs' results are valid.   */
int sba_batch_solve_problem(sba_batch* b, int use_initial_guess, int trials, double subset_fraction, unsigned long long seed,
                            double* rot, double* tran, const sba_lm_options* opt, double* d12_out, double* d_uniform,
                            int* guess_candidates, sba_lm_summary* depth_summaries, sba_lm_summary* rot_summaries,
                            sba_lm_summary* tran_summaries, int* status);

/* The d-only stage (spherical_bundle_adjuster.cpp:196-197, functor :1004-1063) for EVERY pair of the batch: what
 * sba_problem_solve_depths does for one problem, per pair -- its own trust region, projected line search and convergence;
 * the solvers run on the device: the block that owns a pair runs the pair's first passes in one launch, pairs that need more go
 * on in per-pass launches whose blocks are dealt out to the pairs still iterating (SBA_BATCH_DEPTH_FIRST_PASSES=0: the one launch
 * runs every pair to the end; SBA_BATCH_DEVICE_DEPTH=0: host solvers in lock-step, one launch per pass -- these two agree to the
 * bit, the default to 1e-9 with equal counts).  rot, tran: double[num_pairs][3] (frozen);
 * needs per-match depths uploaded (the initial values) and refines them on the device, so that a following
 * SBA_DEPTH_PER_MATCH sweep / solve sees them; d12_out (may be NULL): double[offsets[num_pairs]][2], indexed like the uploaded d12
 * (the reference then takes d12_out[offsets[g]][0] and d12_out[offsets[g] + 1][0] as the pair's uniform depths of the rot /
 * tran stages, .cpp:941-942).  summaries / status as sba_batch_solve.  One 512-thread block per pair: made for
 * batches of many pairs (config C5); a batch of a few huge pairs is served, but by as many CUs as it has pairs.      */
int sba_batch_solve_depths(sba_batch* b, const double* rot, const double* tran, double lambda, double c,
                           const sba_lm_options* opt, double* d12_out, sba_lm_summary* summaries, int* status);

/* ---- callers / data formats either side of the path ------------------------------------- */
/* pixel -> unit sphere (spherical_bundle_adjuster.cpp:271-298).  keypoints: n records of
 * `stride_bytes` bytes whose first two floats are pt.x, pt.y (cv::KeyPoint: stride 28).
 * out_xyz: double[3n] host array (cv::Point3d layout).                                       */
int sba_keypoints_to_sphere(int device, const void* keypoints, size_t n, size_t stride_bytes,
                            int im_width, int im_height, double* out_xyz);

/* The four maps below end in a decision on an f64 expression of sin/cos/acos/atan2 -- an integer pixel index by
 * truncation, or a float32 by rounding.  The device computes every output and marks the few whose f64 value lies within
 * 1e-6 pixel of such a decision boundary (at pitch -90 whole pixel lines sit exactly on one); those are finished on the
 * host by the same source function compiled against the host C library, the one the reference runs on.  Results are
 * therefore bit-identical to the reference's arithmetic, not merely within a pixel.  The *_device forms take device
 * pointers and a HIP stream (NULL = the default stream); they synchronise that stream once (to read the list of
 * marked outputs) -- the image forms only the first time a geometry is seen, its source-index table is cached.     */

/* ERP -> cubemap strip (equi2cube.cpp:12-302).  erp: H x W x 3 bytes (8UC3, row-major, host).
 * out: cube_size x (6*cube_size) x 3 bytes, face order left,front,right,back,top,bottom
 * (equi2cube.cpp:292-298).  Source indices are clamped into the image (the reference does not
 * clamp, equi2cube.cpp:47-50; it can only differ at the exact pole).                        */
int sba_equi2cube(int device, const uint8_t* erp, int im_height, int im_width, int cube_size,
                  uint8_t* out);
/* Batched, device-resident form: erp_dev = batch x H x W x 3 bytes, out_dev = batch x S x 6S x 3
 * bytes, both on `device`; enqueued on `stream`.                                                 */
int sba_equi2cube_device(int device, void* stream, const void* erp_dev, int im_height, int im_width,
                         int cube_size, int batch, void* out_dev);

/* ---- key-point / image maps of the matchers (what turns matcher output into the BA path's input) ------ */
/* spherical_surf::rotate_keypoint (spherical_surf.cpp:110-123): in place on n records of stride_bytes whose
 * first two floats are pt.x, pt.y (band coordinates); adds the band offset H*3/8, rotates by the pitch angle
 * through rotate_pixel (:48-74) with the reference's integer truncations, writes ERP pixel coordinates back. */
int sba_rotate_keypoints(int device, void* keypoints, size_t n, size_t stride_bytes, float pitch_deg,
                         int im_width, int im_height);
int sba_rotate_keypoints_device(int device, void* stream, void* keypoints_dev, size_t n, size_t stride_bytes,
                                float pitch_deg, int im_width, int im_height);
/* equi2cube_surf::cube2equi_pixel (equi2cube_surf.cpp:19-76): in place, cube-strip pixel -> ERP pixel.       */
int sba_cube2equi_keypoints(int device, void* keypoints, size_t n, size_t stride_bytes, int cube_size,
                            int im_width, int im_height);
int sba_cube2equi_keypoints_device(int device, void* stream, void* keypoints_dev, size_t n, size_t stride_bytes,
                                   int cube_size, int im_width, int im_height);
/* spherical_surf::crop_rotated_image (spherical_surf.cpp:76-108): erp H x W x 3 bytes -> out (H/4) x W x 3,
 * the equatorial band of the image rotated by pitch_deg (inverse warping); pixels whose source falls outside
 * the image are 0 (the reference leaves them uninitialised).                                               */
int sba_crop_rotated_image(int device, const uint8_t* erp, int im_height, int im_width, float pitch_deg,
                           uint8_t* out);
/* Batched, device-resident form: erp_dev = batch x H x W x 3, out_dev = batch x (H/4) x W x 3.               */
int sba_crop_rotated_image_device(int device, void* stream, const void* erp_dev, int im_height, int im_width,
                                  float pitch_deg, int batch, void* out_dev);
/* Diagnostics: number of outputs of the cached source-index table that were decided on the host; -1 if that table has
 * not been built.  kind 0 = equi2cube (param = cube_size), 1 = crop (param = the bit pattern of the float pitch).   */
long sba_map_table_host_decided(int device, int kind, int param, int im_height, int im_width);
/* The tiled form of that table, which the gather kernel stages through LDS: number of 32 x 32 output tiles, how many of
 * them are staged (the rest gather from global memory), LDS bytes per frame.  0, or -1 if the table is not built.      */
int sba_map_table_tiles(int device, int kind, int param, int im_height, int im_width, int* tiles, int* staged_tiles,
                        int* lds_bytes_per_frame);

#ifdef __cplusplus
}
#endif
#endif /* SBA_HIP_H_ */
