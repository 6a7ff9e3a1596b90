/*
 * mi355_icp.h — C-ABI of the MI355X-native rigid point-cloud alignment engine.
 *
 * This is the drop-in boundary for ONE path of yumi-crew/pose_estimation: the ICP
 * pose refinement that OpenCVSurfaceMatch::find_object_in_scene runs through
 *     cv::ppf_match_3d::ICP icp(250, 0.005f, 2.5f, 8);        (reference
 *     icp.registerModelToScene(model, scene, poses);           pose_estimation/src/opencv_surface_match.cpp:85,94)
 * Each entry point below names the reference interface it replaces.  Plain
 * pointers and sizes only: no OpenCV, ROS, torch or C++ types cross this line,
 * no exception crosses it either (integer status + mi355_icp_last_error()).
 *
 * Vocabulary follows the reference: the *scene* is the TARGET cloud (searched),
 * the *model* is the SOURCE cloud (moved); a pose is a row-major 4x4 double
 * matrix taking model coordinates to scene coordinates (cv::Matx44d,
 * Pose3D::pose); a 7-float pose is {x,y,z,qx,qy,qz,qw}
 * (reference pose_estimation/include/pose_estimation.hpp:83-88).
 *
 * All compute entry points run on the GPU (gfx950).  There is no CPU fallback:
 * without a usable HIP device mi355_icp_create() fails with
 * MI355_ICP_ERR_NO_DEVICE.  Entry points marked [host] are pure host
 * arithmetic (the 3x3 SVD / convergence / pose packing the design keeps on the
 * CPU) and work without a GPU.
 */
#ifndef MI355_ICP_H
#define MI355_ICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ICP_ABI_VERSION 1

/* Number of double-precision scalars one correspondence pass reduces to.
 * Layout (all sums run over the ACCEPTED correspondences (s0_i, t_j(i))):
 *   [0]      n            number of accepted pairs
 *   [1..3]   sum s0       untransformed source (model) point
 *   [4..6]   sum t        matched target (scene) point
 *   [7..15]  sum s0 t^T   row-major 3x3, row = source component, col = target component
 *   [16]     sum d2       float32 squared distance of each accepted pair
 * [4..15] are the "12 covariance/centroid scalars" of the design brief; [0..3]
 * and [16] are the bookkeeping the convergence test and rejection need.  This
 * block is what crosses xGMI once per iteration in a slab-sharded run (all-gathered, then added in rank order). */
#define MI355_ICP_NSUMS 17

typedef enum mi355_icp_status {
  MI355_ICP_OK = 0,
  MI355_ICP_ERR_INVALID_ARGUMENT = 1,
  MI355_ICP_ERR_NO_DEVICE = 2,        /* no HIP device / HIP runtime failure at create */
  MI355_ICP_ERR_HIP = 3,              /* a HIP call failed; see mi355_icp_last_error */
  MI355_ICP_ERR_NONFINITE_INPUT = 4,  /* NaN/Inf in an uploaded cloud or pose (the reference strips NaNs
                                         before this path: pose_estimation.cpp:248) */
  MI355_ICP_ERR_TOO_FEW_PAIRS = 5,    /* < 3 accepted correspondences: pose left unrefined */
  MI355_ICP_ERR_NUMERIC = 6,          /* non-finite sums / degenerate solve */
  MI355_ICP_ERR_COMM = 7,             /* RCCL failure or RCCL not loadable */
  MI355_ICP_ERR_INTERNAL = 8
} mi355_icp_status;

typedef struct mi355_icp_ctx mi355_icp_ctx; /* opaque; one per OpenCVSurfaceMatch instance
                                               (reference pose_estimation.hpp:161) */

/* Alignment parameters.  The numeric definition (exact float32 NN, double sums,
 * Kabsch) is fixed; these only steer the iteration. */
typedef struct mi355_icp_params {
  int32_t max_iterations;              /* >= 1.  Reference call site passes 250 (opencv_surface_match.cpp:85) */
  float   max_correspondence_distance; /* pairs with d2 > this^2 (float32) are rejected; <= 0 => keep all */
  double  transformation_epsilon;      /* criterion A: converged when |dt|^2 <= this AND 1-cos(dtheta) <= rotation_epsilon
                                          (incremental transform of one iteration); <= 0 disables A */
  double  rotation_epsilon;            /* used with criterion A */
  double  relative_mse_epsilon;        /* criterion B: converged when |mse_k - mse_{k-1}| <= this * mse_{k-1};
                                          <= 0 disables B.  Reference passes tolerance 0.005 (same call site) */
  int32_t flags;                       /* MI355_ICP_FLAG_* */
} mi355_icp_params;

#define MI355_ICP_FLAG_PROFILE 1  /* bracket every kernel launch with HIP events (see mi355_icp_get_profile) */
#define MI355_ICP_FLAG_PRUNED_SCAN 2 /* answer the nearest-neighbour queries with the exact PRUNED scan instead of the
                                        brute-force one: identical indices, distances and poses from a fraction of the
                                        distance evaluations (Morton-sorted copies of both clouds, bounding spheres per
                                        tile and run).  Off by default: the reference path this engine replaces is
                                        specified - and benchmarked - as brute force.  Worth it from ~100k points. */

#define MI355_ICP_FLAG_BRUTE_SCAN 8  /* mi355_icp_align_reference only: every pass by the brute-force scan.  Without this flag and
                                        without MI355_ICP_FLAG_PRUNED_SCAN that call picks, per pass, whichever of the two
                                        exact scans is estimated faster (the pruned one from a few hundred thousand scene
                                        points on; same indices, distances and poses either way; MI355_ICP_SERVICE_PRUNE=0|1
                                        overrides the estimate).  mi355_icp_align - the benchmarked path - is brute force
                                        unless MI355_ICP_FLAG_PRUNED_SCAN asks otherwise. */
#define MI355_ICP_FLAG_UPSTREAM 4    /* mi355_icp_align_reference only: follow upstream cv::ppf_match_3d::ICP where the round 1-3
                                        form of reference mode left its control flow (see mi355_icp_reference_params).
                                        The host mirrors (mi355::cv_adapter::ICP, mi355::ppf_match_3d::ICP, Python ICP) set
                                        it by default; needs the MODEL uploaded with normals too (N x 6 rows). */

/* Kernel timing collected when MI355_ICP_FLAG_PROFILE is set; cumulative since
 * the last mi355_icp_reset_profile(). */
typedef struct mi355_icp_profile {
  double   nn_scan_ms;       /* brute-force NN scan kernel (the dominant kernel) */
  double   resolve_ms;       /* index resolve + covariance accumulation kernel */
  double   reduce_ms;        /* fixed-order final reduction kernel */
  uint64_t nn_scan_launches;
  uint64_t resolve_launches;
  uint64_t reduce_launches;
  double   pairs_evaluated;  /* sum over nn_scan launches of Ns*Nt actually scanned */
  uint64_t iterations;       /* ICP iterations (correspondence passes) executed */
  double   knn_scan_ms;      /* scene normals: k-NN scan kernel */
  double   normals_ms;       /* scene normals: merge + PCA kernel */
  uint64_t normals_launches;
  uint64_t overflow_points;  /* filter scans: source points handed to the all-exact scan (>= 3 candidate chunks) */
  double   scene_filter_ms;  /* scene compaction kernel */
  uint64_t scene_filter_launches;
  uint64_t scene_points_in;  /* points read / written by those launches */
  uint64_t scene_points_out;
  uint64_t overflow_host_passes; /* passes whose overflow points went through the host path (gather -> all-exact tiled
                                    scan -> scatter) instead of the device-side pass: more than 1024 of them in one job,
                                    a large target, or a communicator attached */
} mi355_icp_profile;

/* ---- lifetime ------------------------------------------------------------ */

unsigned mi355_icp_abi_version(void); /* [host] */

/* Creates an engine bound to HIP device `device_id` with its own stream.
 * Replaces: construction of cv::ppf_match_3d::ICP (opencv_surface_match.cpp:85). */
int mi355_icp_create(int device_id, mi355_icp_ctx** out);
void mi355_icp_destroy(mi355_icp_ctx* ctx);

/* ONE handle over `n_devices` devices of THIS process (SURVEY.md 8b "Lifetime" / "Threading": the reference holds its backend
 * by value in one node process - pose_estimation.hpp:160-162, one component container: launch/pose_estimation.launch.py:17-35 -
 * so multi-GPU must not need a launcher or a rank per process).  The library keeps one engine, one stream and one host thread
 * per entry of device_ids (an id may repeat: two engines on one device, which is how the one-GPU test boxes run it); every
 * call below works on the handle unchanged and is fanned out internally:
 *   - clouds: the scene (targets) is replicated on every device; the models (sources) are kept on the host as uploaded and
 *     laid out per call;
 *   - mi355_icp_align / _align_reference / _correspond with AT LEAST as many jobs as devices (batched bin-picking: BASELINE
 *     config[3]; a request's candidates on two GPUs): whole jobs per device, no exchange, every job's result is the
 *     single-device engine's bit for bit;
 *   - with FEWER jobs than devices (BASELINE config[4], one 8M-point model): every device takes a slab of each source cloud
 *     (mi355_icp_slab_range) and the per-pass rows of MI355_ICP_NSUMS (MI355_ICP_NREFSUMS_EX) doubles are exchanged through
 *     pinned host memory and added in device order - no RCCL, no torch; the arithmetic of a slab-sharded run of the same
 *     rank count (mi355_icp_comm_attach*), bit for bit;
 *   - scene preparation (mi355_icp_scene_*) runs on every device (each needs the whole scene; the kernels are deterministic);
 *     host outputs come from device slot 0, as do mi355_icp_get_profile and mi355_icp_device_info;
 *   - mi355_icp_comm_attach* are refused (the handle shards by itself); mi355_icp_comm_info reports kind 3, n_devices ranks.
 * Calls on one handle must still come from one thread at a time (the reference's executor is single-threaded). */
int mi355_icp_create_multi(const int* device_ids, int n_devices, mi355_icp_ctx** out);

/* Last error text of this context (never NULL; "" if none).  With ctx == NULL
 * returns the text of the last failed mi355_icp_create on this thread. */
const char* mi355_icp_last_error(const mi355_icp_ctx* ctx); /* [host] */

/* Fills the defaults of the design: 50 iterations, no rejection, criteria A and B off
 * (rotation_epsilon preset to 1e-5 for when A is switched on).  [host] */
void mi355_icp_default_params(mi355_icp_params* p);

/* Maps the reference's constructor arguments ICP(iterations, tolerance,
 * rejectionScale, numLevels) (opencv_surface_match.cpp:85) onto params.
 * iterations -> max_iterations, tolerance -> relative_mse_epsilon, for the POINT-TO-POINT loop
 * (mi355_icp_align), which has no use for rejectionScale and numLevels: they are validated and dropped
 * here.  All four arguments are live in reference mode - mi355_icp_reference_params /
 * mi355_icp_align_reference below - which is what the reference's call really runs.  [host] */
int mi355_icp_params_from_reference(int iterations, float tolerance, float rejection_scale, int num_levels,
                                    mi355_icp_params* p);

/* ---- clouds -------------------------------------------------------------- */

/* Uploads `n_clouds` target (scene) clouds, replacing any previous targets.
 * xyz[c] points at rows of >= 3 float32 (x,y,z first) `stride_bytes` apart: 12
 * for the M x 3 scene (reference pose_estimation.cpp:263-276), 24 for M x 6
 * with normals (opencv_surface_match.cpp:57-59).  The library copies and
 * re-lays the data out as SoA on the device; the caller's buffer may be freed
 * on return.  Replaces: the `scene` argument of registerModelToScene. */
int mi355_icp_set_targets(mi355_icp_ctx* ctx, int n_clouds, const float* const* xyz, const size_t* n_points,
                          size_t stride_bytes);

/* Same for source (model) clouds; these persist across requests like
 * OpenCVSurfaceMatch::models_ (opencv_surface_match.hpp:45).  Replaces: the
 * `model` argument of registerModelToScene.
 * For slab-sharded multi-GPU runs each rank uploads only its slab of the
 * source (see mi355_icp_slab_range). */
int mi355_icp_set_sources(mi355_icp_ctx* ctx, int n_clouds, const float* const* xyz, const size_t* n_points,
                          size_t stride_bytes);

/* Contiguous slab [begin, begin+count) of `n` points owned by `rank` of
 * `n_ranks` (sizes differ by at most one, lower ranks get the extra).  [host] */
int mi355_icp_slab_range(size_t n, int rank, int n_ranks, size_t* begin, size_t* count);

/* Slab-sharded reference mode: which points of the slab [begin, begin + count) of a cloud of n_total points belong to
 * pyramid level `level` of the WHOLE cloud (level l keeps the global indices 0, 2^l, 2 * 2^l, ... below
 * (n_total >> l) << l, as the single-GPU pyramid does): offset of the first one inside the slab, how many there are,
 * the level index of the first, and the size of the whole level.  Any output may be NULL.  [host] */
int mi355_icp_level_slab(size_t n_total, size_t begin, size_t count, int level, size_t* first_local, size_t* n_level_local,
                         size_t* level_index_base, size_t* n_level_total);
/* The same for a level that keeps every step-th point (global indices 0, step, 2 step, ..., n_total / step of them):
 * the upstream-faithful variant samples with upstream's step, which is 2^level only for large models.  [host] */
int mi355_icp_level_slab_step(size_t n_total, size_t begin, size_t count, size_t step, size_t* first_local,
                              size_t* n_level_local, size_t* level_index_base, size_t* n_level_total);
/* Upstream's sampling step of pyramid level `level` for a model of n_model rows: numSamples = cvRound(n / 2^level),
 * sampleStep = cvRound(n / numSamples), round half to even (recalled from upstream; unverifiable offline).  Returns 0 when
 * the level has no samples.  [host] */
unsigned mi355_icp_upstream_step(size_t n_model, int level);

/* ---- alignment ----------------------------------------------------------- */

/* Refines `n_jobs` poses in one batched call.  Job k aligns source cloud
 * source_ids[k] onto target cloud target_ids[k] starting from the row-major
 * 4x4 double pose init_poses[16*k..] (model -> scene).  The reference batches
 * <= 6 candidate poses of one model against one scene
 * (opencv_surface_match.cpp:89-94); any mix of clouds is allowed here.
 *
 * Outputs (caller-allocated, may be NULL to skip):
 *   out_poses      16*n_jobs doubles, refined pose  (Pose3D::pose)
 *   out_residual   n_jobs doubles, RMS correspondence distance of the last
 *                  correspondence pass               (Pose3D::residual)
 *   out_iterations n_jobs ints, correspondence passes executed
 *   out_status     n_jobs ints: MI355_ICP_OK (converged or ran max_iterations;
 *                  see out_converged), MI355_ICP_ERR_TOO_FEW_PAIRS or
 *                  MI355_ICP_ERR_NUMERIC, in which case out_poses[k] is the
 *                  last valid pose (the initial pose if the first pass failed):
 *                  "leave the candidate unrefined", matching how the reference
 *                  swallows exceptions (opencv_surface_match.cpp:144-147)
 *   out_converged  n_jobs ints, 1 if criterion A or B fired
 * Returns MI355_ICP_OK unless the call as a whole could not run.
 * With a communicator attached every rank must make the same call (same jobs,
 * same poses); the ranks' sums are exchanged and added in rank order so every rank returns identical poses.
 * Replaces: ICP::registerModelToScene(model, scene, poses). */
int mi355_icp_align(mi355_icp_ctx* ctx, int n_jobs, const int* source_ids, const int* target_ids,
                    const double* init_poses, const mi355_icp_params* params, double* out_poses,
                    double* out_residual, int* out_iterations, int* out_status, int* out_converged);

/* One correspondence pass (no solve), for parity checks and for drivers that
 * run their own collective.  For each job: transform the source by the float32
 * rounding of poses[16*k..], exact 1-NN, accumulate.
 *   out_index   concatenated over jobs, source-cloud order: int32 target index  (may be NULL)
 *   out_dist2   same shape: float32 squared distance                         (may be NULL)
 *   out_sums    MI355_ICP_NSUMS * n_jobs doubles, LOCAL to this rank (never exchanged)
 * max_correspondence_distance as in params (<= 0 keeps all).
 * A source point whose distance to EVERY target is +inf or NaN (coordinates or a pose that overflow float32)
 * gets the oracle's answer for "nothing below +inf": index 0, d2 = +inf; the call still returns MI355_ICP_OK and
 * mi355_icp_last_error() says that it happened.  (In mi355_icp_align such a pair makes the job's sums non-finite
 * and the job ends with MI355_ICP_ERR_NUMERIC, its pose left at the last valid value.) */
int mi355_icp_correspond(mi355_icp_ctx* ctx, int n_jobs, const int* source_ids, const int* target_ids,
                         const double* poses, float max_correspondence_distance, int32_t* out_index,
                         float* out_dist2, double* out_sums);

/* Diagnostic for the parity tests (no reference counterpart; no product path calls it): the lower bounds the
 * matrix-core filter scan works with.  Runs the production scan kernel (same code, instantiated with a dump switch)
 * for source cloud `source_id` under `pose` against target cloud `target_id` and returns, for EVERY source point and
 * EVERY run of 128 consecutive targets of the Morton-sorted copy, the lower bound L the scan computed for the exact
 * float32 minimum of that run (0 where the scan makes no statement: non-finite bound).
 *   out_lower        n_runs x n_source floats, run-major: out_lower[g * n_source + i], i in upload order
 *   out_run_targets  n_runs x 128 int32: the original indices of run g's targets, -1 for padding slots
 *   inout_n_runs     in: capacity of the buffers in runs; out: number of runs (padded target count / 128)
 * Fails with MI355_ICP_ERR_INVALID_ARGUMENT if the capacity is too small (inout_n_runs still receives the count). */
int mi355_icp_debug_filter_bounds(mi355_icp_ctx* ctx, int source_id, int target_id, const double* pose, float* out_lower,
                                  int32_t* out_run_targets, size_t* inout_n_runs);

/* ---- reference mode: what cv::ppf_match_3d::ICP really runs ---------- */

/* The constructor arguments of the reference call site, cv::ppf_match_3d::ICP(iterations, tolerance,
 * rejectionScale, numLevels) (opencv_surface_match.cpp:85), all live in this mode:
 * resolution pyramid (level l subsamples both clouds with step 2^l, at most round(iterations/(l+1))
 * passes, stop when fval/fval_old is inside 1 +- tolerance*(l+1)^2), exact 1-NN, median + MAD
 * rejection (rejection_scale <= 0 keeps all), "picky" one-to-one dedupe, linearised point-to-plane
 * least squares on the TARGET (scene) normals.  Numeric definition: oracle/icp_oracle.c, second half.
 * Needs targets uploaded as rows of >= 6 float32 (xyz + normal, stride_bytes >= 24). */
typedef struct mi355_icp_reference_params {
  int32_t iterations;      /* reference passes 250 */
  float   tolerance;       /* reference passes 0.005f */
  float   rejection_scale; /* reference passes 2.5f */
  int32_t num_levels;      /* reference passes 8 */
  int32_t flags;           /* MI355_ICP_FLAG_* */
} mi355_icp_reference_params;

#define MI355_ICP_NREFSUMS 29    /* 21 upper-triangle A^T A (row-major), 6 A^T b, sum |p-q|^2, number of pairs */
#define MI355_ICP_NREFSUMS_EX 30 /* those, then sum |n_p - n_q|^2 over the selected pairs (MI355_ICP_FLAG_UPSTREAM; else 0) */

/* Batched like mi355_icp_align.  out_passes[k] is the number of correspondence passes over all levels.  A job whose
 * level ends early (< 6 pairs, degenerate solve) keeps what it has, as upstream does.
 *
 * Two forms (numeric definition of both: oracle/icp_oracle.c, second half; the differences: INTEGRATION.md 4a):
 *  - flags without MI355_ICP_FLAG_UPSTREAM (rounds 1-3): level step 2^l, plane equations about the cloud origin,
 *    out_residual[k] = smallest fval = sqrt(sum |p-q|^2) / n_level of the finest level that ran, in cloud units;
 *  - MI355_ICP_FLAG_UPSTREAM (what a cv::ppf_match_3d::ICP built from upstream's source does, as recalled): the model's
 *    normals travel with it (the model must be uploaded as N x 6 rows, as the reference's is: opencv_surface_match.cpp:22),
 *    upstream's pre-normalisation - joint-mean centre, scale n / mean distance - is applied algebraically (the plane
 *    equations are written about that centre, the pose is composed in that frame and returned in cloud coordinates),
 *    fval is the norm over all six columns of the matched rows in the normalised units, so out_residual[k] is the number
 *    upstream's Pose3D::residual holds, the level step is cvRound(n / cvRound(n / 2^l)) of the MODEL size for both clouds,
 *    and the pass limit rounds half to even.
 * Replaces: ICP::registerModelToScene(model, scene, poses) with the reference's parameters. */
int mi355_icp_align_reference(mi355_icp_ctx* ctx, int n_jobs, const int* source_ids, const int* target_ids,
                              const double* init_poses, const mi355_icp_reference_params* params, double* out_poses,
                              double* out_residual, int* out_passes, int* out_status);

/* One reference-mode pass on pyramid level `level` of (source_id, target_id), for parity checks:
 * pose_start defines the point the plane equations are written for, pose_cur the pose the NN search runs under.
 * Outputs are level-local (n_source >> level entries): target index, float32 d2, selected mask (after rejection
 * and picky), the rejection threshold, and the MI355_ICP_NREFSUMS sums.  Any output may be NULL. */
int mi355_icp_reference_pass(mi355_icp_ctx* ctx, int source_id, int target_id, int level, const double* pose_start,
                             const double* pose_cur, float rejection_scale, int32_t* out_index, float* out_dist2,
                             uint8_t* out_selected, float* out_threshold, double* out_sums);

/* One pass of either form on the level that keeps every step-th point of both clouds (step >= 1; upstream's steps are
 * not always powers of two), for parity checks.  centre == NULL: the round 1-3 form (out_sums[29] = 0); centre given
 * (3 floats): the upstream-faithful form - plane equations about centre, out_sums[29] = sum |n_p - n_q|^2 - which needs the
 * source uploaded with normals.  Outputs are level-local (n_source / step entries); out_sums holds MI355_ICP_NREFSUMS_EX
 * doubles.  Any output may be NULL. */
int mi355_icp_reference_pass_ex(mi355_icp_ctx* ctx, int source_id, int target_id, unsigned step, const double* pose_start,
                                const double* pose_cur, float rejection_scale, const float* centre, int32_t* out_index,
                                float* out_dist2, uint8_t* out_selected, float* out_threshold, double* out_sums);

/* Centre and scale of upstream's pre-normalisation for (source_id under `pose`, target_id), as MI355_ICP_FLAG_UPSTREAM
 * computes them (for parity checks): out_centre 3 floats, out_scale one double. */
int mi355_icp_upstream_normalisation(mi355_icp_ctx* ctx, int source_id, int target_id, const double* pose, float* out_centre,
                                     double* out_scale);

/* Minimum-norm least-squares solution x = (rx, ry, rz, tx, ty, tz) of the 6x6 normal equations packed in
 * sums (symmetric eigen-decomposition in double).  [host] */
int mi355_icp_solve_point_to_plane(const double* sums29, double out_x[6]);
/* The same for n systems (sums29: n x MI355_ICP_NREFSUMS, out_x: n x 6, out_status: n x MI355_ICP_OK or
 * MI355_ICP_ERR_NUMERIC), eight per SIMD pass - what a reference-mode pass uses for its candidates.  Every system gets
 * the bits mi355_icp_solve_point_to_plane would give it.  [host] */
int mi355_icp_solve_point_to_plane_batch(const double* sums29, int n, double* out_x, int* out_status);

/* ---- scene preparation: the reference's point filters as one-pass stream compaction ---- */

/* The filters PoseEstimation::create_surface_match_pc applies before the ICP sees the scene
 * (reference pose_estimation/src/pose_estimation.cpp): NaN removal (:248), the sphere filter filter_points
 * (:347-372) and the distance band around a fitted plane in remove_planes (:310-333).  Arithmetic follows those
 * lines (float32, their operation order, the band compared in double); kept points stay in their original order
 * (the reference's loops emit indices in thread-arrival order under `omp critical`).  The plane comes from
 * mi355_icp_scene_fit_plane below (the replacement of the pcl::SACSegmentation call at :285-297) or from the caller. */
typedef enum mi355_scene_filter_kind {
  MI355_SCENE_FILTER_FINITE = 0,               /* keep points whose x, y, z are all finite */
  MI355_SCENE_FILTER_SPHERE_KEEP_INSIDE = 1,   /* p[0..2] centre, p[3] radius: keep d <= radius  (filter_out != "inliers") */
  MI355_SCENE_FILTER_SPHERE_REMOVE_INSIDE = 2, /* same test, keep d > radius                      (filter_out == "inliers") */
  MI355_SCENE_FILTER_PLANE_BAND_REMOVE = 3     /* p[0..3] = a, b, c, d of ax+by+cz+d=0, p[4] = band (reference: 0.005):
                                                  remove |(x*a+y*b+z*c+d)/|p|| <= band */
} mi355_scene_filter_kind;

typedef struct mi355_scene_filter {
  int32_t kind;
  float   p[5];
} mi355_scene_filter;

#define MI355_SCENE_MAX_FILTERS 8

/* Copies a raw scene (rows of >= 3 float32, stride_bytes apart; NaN/Inf allowed: an organised camera frame has
 * them) to the GPU as the current scene.  Replaces: the PointCloud2 -> pcl cloud copy (pose_estimation.cpp:229-244). */
int mi355_icp_scene_load(mi355_icp_ctx* ctx, const float* xyz, size_t n_points, size_t stride_bytes);

/* Applies the AND of `n_filters` (<= MI355_SCENE_MAX_FILTERS) predicates in ONE pass over the current scene and
 * compacts it on the device; *out_n_points (optional) receives the number of points kept. */
int mi355_icp_scene_filter(mi355_icp_ctx* ctx, int n_filters, const mi355_scene_filter* filters, size_t* out_n_points);

/* Copies the current scene back as packed xyz rows (for the host-side RANSAC between plane removals).
 * Replaces: the pcl cloud -> cv::Mat N x 3 copy (pose_estimation.cpp:263-276). */
int mi355_icp_scene_download(mi355_icp_ctx* ctx, float* out_xyz, size_t capacity_points, size_t* out_n_points);

/* RANSAC plane fit of the current scene ("RANSAC plane fit ... becomes a small batched kernel").
 * Replaces: the pcl::SACSegmentation call of PoseEstimation::remove_planes (reference pose_estimation.cpp:285-297:
 * SACMODEL_PLANE, SAC_RANSAC, setOptimizeCoefficients(true), setDistanceThreshold(0.0001), setMaxIterations(100)), whose
 * coefficients then feed MI355_SCENE_FILTER_PLANE_BAND_REMOVE (:310-333).  PCL's structure, restated (its source is not
 * in the reference, its random sample sequence is not reproducible: the fitted plane agrees with PCL's only to within
 * what two RANSAC runs of PCL agree with each other - parity unpinned; oracle/icp_oracle.c pins THIS definition):
 *   hypotheses  h = 0 .. max_iterations-1: three distinct points drawn by a counter-based generator keyed by `seed`
 *               (up to 16 attempts per hypothesis for a non-degenerate triple); plane through them, float32:
 *               n = (p1-p0) x (p2-p0) normalised, d = -n.p0
 *   support     number of points with |a x + b y + c z + d| <= distance_threshold (float32), all hypotheses in ONE pass
 *               over the scene
 *   choice      PCL's adaptive rule replayed in hypothesis order: a hypothesis with more support than any before it
 *               becomes the best and sets k = log(1 - probability) / log(1 - w^3), w = support / points; the search
 *               ends with the first h + 1 >= k
 *   refinement  (optimize != 0) least squares on the best hypothesis's inliers: centroid and covariance in double,
 *               eigenvector of the smallest eigenvalue, sign kept towards the hypothesis; support is counted again
 * Non-finite points never support a plane.  out_coefficients: a, b, c, d of ax + by + cz + d = 0 with |(a,b,c)| = 1. */
typedef struct mi355_plane_fit_params {
  float    distance_threshold; /* reference: 0.0001 */
  int32_t  max_iterations;     /* reference: 100; 1 .. 1024 */
  double   probability;        /* PCL default 0.99 */
  uint64_t seed;
  int32_t  optimize;           /* reference: setOptimizeCoefficients(true) */
  int32_t  reserved;
} mi355_plane_fit_params;

int mi355_icp_scene_fit_plane(mi355_icp_ctx* ctx, const mi355_plane_fit_params* params, float out_coefficients[4],
                              size_t* out_inliers, int* out_iterations);

/* Normals of the current scene by k-nearest-neighbour PCA.
 * Replaces: cv::ppf_match_3d::computeNormalsPC3d(pc_scene, pc_scene_normals, 20, true, viewpoint)
 * (opencv_surface_match.cpp:57-59): for every point its k nearest points of the scene (itself included, exact
 * float32 distances, (d2, index) order), covariance accumulated as upstream does (float32 products in double),
 * eigenvector of the smallest eigenvalue, flipped (flip_viewpoint != 0) so that (viewpoint - point) . n >= 0.
 * 1 <= k <= 32 (clamped to the number of points).  The scene must be finite.  out_normals (n x 3 float32) and
 * out_neighbors (n x min(k, n) int32, neighbour indices in (d2, index) order) are optional host buffers; the normals
 * stay on the device and are adopted by mi355_icp_scene_commit_target, which makes reference mode usable on a
 * scene that arrived without normals. */
int mi355_icp_scene_normals(mi355_icp_ctx* ctx, int k, int flip_viewpoint, const float viewpoint[3], float* out_normals,
                            int32_t* out_neighbors);

/* Makes the current scene target cloud 0 (replacing all targets) without leaving the device, together with its
 * normals if mi355_icp_scene_normals ran on it.  Fails with MI355_ICP_ERR_NONFINITE_INPUT if the scene still holds
 * NaN/Inf. */
int mi355_icp_scene_commit_target(mi355_icp_ctx* ctx);

/* ---- host arithmetic (no GPU needed) ------------------------------------- */

/* Kabsch: the rigid transform minimising sum |R s0 + t - t_j|^2 over the pairs
 * summarised by `sums` (3x3 SVD in double, reflection fixed).  Writes a
 * row-major 4x4.  Returns MI355_ICP_ERR_TOO_FEW_PAIRS if n < 3,
 * MI355_ICP_ERR_NUMERIC on non-finite input.  [host] */
int mi355_icp_solve_rigid(const double sums[MI355_ICP_NSUMS], double out_pose[16]);

/* One step of the host iteration controller used by mi355_icp_align (exposed
 * so an external driver - e.g. one all-reducing `sums` itself - follows the
 * identical rule).  `iteration` is the 0-based index of the pass that produced
 * `sums`; `pose` is the pose that pass was evaluated under and is overwritten
 * with the new pose; `prev_mse` carries mse_{k-1} (ignored when iteration==0)
 * and is overwritten with mse_k; *out_done = 1 when iteration+1 reached
 * max_iterations or a criterion fired, *out_converged = 1 only for the latter.
 * On a non-OK return pose is left untouched.  [host] */
int mi355_icp_host_update(const mi355_icp_params* params, const double sums[MI355_ICP_NSUMS], int iteration,
                          double pose[16], double* prev_mse, int* out_done, int* out_converged);

/* Row-major 4x4 pose -> 7 floats {x,y,z,qx,qy,qz,qw} using the reference's own
 * matrix->quaternion formula (chessboard_pose_estimator.cpp:196-207).
 * reference_packing != 0 reproduces the packing defect of
 * opencv_surface_match.cpp:133-142 byte for byte (out[5] left 0, out[6] = qw);
 * 0 gives the intended full quaternion.  [host] */
int mi355_icp_pose_to_pose7(const double pose[16], float out7[7], int reference_packing);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ---------------------- */

#define MI355_ICP_UNIQUE_ID_BYTES 128

/* Rank 0 obtains an RCCL unique id (ncclGetUniqueId) and ships the bytes to the
 * other ranks by any means (bench.py uses torch.distributed).  [host+RCCL] */
int mi355_icp_comm_unique_id(unsigned char out_id[MI355_ICP_UNIQUE_ID_BYTES]);

/* Joins this context to an RCCL communicator of `n_ranks` ranks.  Afterwards mi355_icp_align and
 * mi355_icp_align_reference run slab-sharded: every rank makes the same call on its own slab of each source cloud
 * and receives identical results.  Per correspondence pass the MI355_ICP_NSUMS (or MI355_ICP_NREFSUMS) doubles of
 * every job are exchanged ONCE: the per-rank rows are all-gathered (ncclAllGather on the engine's stream) and added
 * in rank order on every rank, so the sums - hence poses and convergence decisions - are bit-identical on all ranks
 * and do not depend on the collective algorithm RCCL picks.  Reference mode adds integer
 * collectives, exact in any order: six ncclAllReduce(sum, uint32) of a 2048-bin histogram per pass for the MAD
 * threshold, and for the picky dedupe either one ncclAllReduce(min, uint64) of the owner table or - when it is fewer bytes -
 * one ncclAllGather of the (key, slot) pairs each rank offers, applied locally with atomicMin (the same table either way). */
int mi355_icp_comm_attach(mi355_icp_ctx* ctx, const unsigned char id[MI355_ICP_UNIQUE_ID_BYTES], int rank,
                          int n_ranks);

/* The same exchange over a transport the CALLER supplies: `allgather(user, send, recv, bytes_per_rank)` must place
 * rank r's `bytes_per_rank` bytes at recv + r * bytes_per_rank on every rank (host memory) and return 0.  The engine
 * stages through pinned host buffers and performs every reduction itself, in rank order.  Used where RCCL cannot
 * run (ranks sharing one GPU, gloo in the tests); slower than the native path by one host synchronisation per
 * exchange. */
typedef int (*mi355_icp_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
int mi355_icp_comm_attach_external(mi355_icp_ctx* ctx, int rank, int n_ranks, mi355_icp_allgather_fn allgather,
                                   void* user);
int mi355_icp_comm_detach(mi355_icp_ctx* ctx);

/* What is attached: *kind = 0 none, 1 RCCL (rank and size as the communicator itself reports them:
 * ncclCommUserRank / ncclCommCount), 2 external, 3 a multi-device handle (mi355_icp_create_multi: n_ranks = its devices).
 * Any output may be NULL. */
int mi355_icp_comm_info(const mi355_icp_ctx* ctx, int* rank, int* n_ranks, int* kind);

/* ---- measurement ---------------------------------------------------------- */

/* Turns event timing on (1) or off (0) for EVERY pass of this context, including mi355_icp_correspond
 * (mi355_icp_align additionally honours MI355_ICP_FLAG_PROFILE per call). */
int mi355_icp_set_profiling(mi355_icp_ctx* ctx, int enabled);
int mi355_icp_get_profile(const mi355_icp_ctx* ctx, mi355_icp_profile* out);
int mi355_icp_reset_profile(mi355_icp_ctx* ctx);

/* Device name / CU count of the bound device (name buffer >= 64 bytes). */
int mi355_icp_device_info(const mi355_icp_ctx* ctx, char* name, size_t name_len, int* compute_units,
                          size_t* total_mem);

/* ---- synthetic clouds (bench and tests; exact IEEE +,-,*,/,sqrt only, so the
 * output is bit-reproducible on any host) ---------------------------------- */

/* Writes `count` points, index first_index.., of the synthetic "bumpy sphere"
 * surface sample with PRNG key `seed` as AoS float32 xyz.  If sigma > 0 adds
 * zero-mean noise (unit-variance Irwin-Hall(4) * sigma) keyed by noise_seed.
 * If transform12 != NULL applies the row-major 3x4 double rigid transform
 * before rounding to float32.  [host] */
int mi355_icp_synth_cloud(uint64_t seed, uint64_t noise_seed, double sigma, const double* transform12,
                          size_t first_index, size_t count, float* out_xyz);

/* Same sample as rows of 6 float32: xyz then the unit outward normal of the noise-free surface (rotated
 * with the transform) - the N x 6 layout of the reference's clouds with normals
 * (opencv_surface_match.cpp:22,57-59).  [host] */
int mi355_icp_synth_cloud6(uint64_t seed, uint64_t noise_seed, double sigma, const double* transform12,
                           size_t first_index, size_t count, float* out_xyz_normal);

#ifdef __cplusplus
}
#endif
#endif /* MI355_ICP_H */
