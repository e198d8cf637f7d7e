/* dvp_mvs.h — C ABI of the MI355X PatchMatch engine (libdvp_mvs_hip.so).
 *
 * Drop-in boundary for the per-view depth/normal path of ZhenlongYuan/DVP-MVS: these entry
 * points are what the reference's `class APD` methods (APD.h:94-199) do with the CUDA runtime
 * — allocate/upload (APD::CudaSpaceInitialization, APD.cpp:1497-1613), bundle the buffers
 * (APD::SetDataPassHelperInCuda, APD.cpp:1670-1704; DataPassHelper, APD.h:60-92), run the kernel
 * sequence (APD::RunPatchMatch, APD.cu:4406-4532) and copy the results back (APD.cu:4525-4530).
 * Plain pointers and sizes only; POD layouts are the reference's (main.h:58-67, 86-112).
 * All functions return 0 on success, non-zero on error (dvp_last_error() gives the text); the
 * C++ mirror of `class APD` (dvp-mvs_amd/host/APD.h) turns non-zero into the reference's
 * print-and-exit (CudaSafeCall, APD.cpp:943-951).
 * One context == one reference view on one device/stream; contexts are independent (no global
 * state), so one process can own several and N processes can own one GPU each.
 */
#ifndef DVP_MVS_H_
#define DVP_MVS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVP_MAX_IMAGES 32        /* main.h:39 */
#define DVP_NEIGHBOUR_NUM 12     /* main.h:40 */
#define DVP_EDGE_NEIGH_NUM 8     /* main.h:43 */
#define DVP_LAB_BOUNDARY_NUM 8   /* main.h:44 */

/* struct Camera, main.h:58-67 (112 bytes). x_cam = R X + t, c = -R^T t. */
typedef struct DvpCamera {
	float K[9];
	float R[9];
	float t[3];
	float c[3];
	int32_t height;
	int32_t width;
	float depth_min;
	float depth_max;
} DvpCamera;

/* enum RunState, main.h:74-78 ; enum PixelState, main.h:80-84 */
enum { DVP_FIRST_INIT = 0, DVP_REFINE_INIT = 1, DVP_REFINE_ITER = 2 };
enum { DVP_WEAK = 0, DVP_STRONG = 1, DVP_UNKNOWN = 2 };

/* struct PatchMatchParams, main.h:86-112 (76 bytes; C++ bool == uint8_t) */
typedef struct DvpParams {
	int32_t max_iterations;
	int32_t num_images;
	float sigma_spatial;
	float sigma_color;
	int32_t top_k;
	float depth_min;
	float depth_max;
	uint8_t geom_consistency;
	int32_t strong_radius;
	int32_t strong_increment;
	int32_t weak_radius;
	int32_t weak_increment;
	uint8_t use_APD;
	uint8_t use_edge;
	uint8_t use_limit;
	uint8_t use_label;
	uint8_t use_detail;
	uint8_t use_radius;
	int32_t weak_peak_radius;
	int32_t rotate_time;
	float ransac_threshold;
	float geom_factor;
	int32_t state;
} DvpParams;

typedef struct dvp_ctx dvp_ctx;

/* Kernel launches of APD::RunPatchMatch, one id per launch site (APD.cu:4430-4505). */
enum {
	DVP_ST_GEN_EDGE_INFORM = 0,      /* GenEdgeInform          APD.cu:4433 */
	DVP_ST_FIND_NEAREST_STRONG = 1,  /* FindNearestStrongPoint APD.cu:4445 */
	DVP_ST_GEN_NEIGHBOURS = 2,       /* GenNeighbours          APD.cu:4448 */
	DVP_ST_NEIGHBOUR_UPDATE = 3,     /* NeigbourUpdate         APD.cu:4451 */
	DVP_ST_RANDOM_INIT = 4,          /* RandomInitialization   APD.cu:4475 */
	DVP_ST_STRONG_UPDATE = 5,        /* Black/RedPixelUpdateStrong  APD.cu:4479-4481 */
	DVP_ST_RANSAC_FIT = 6,           /* RANSACToGetFitPlane    APD.cu:4484 */
	DVP_ST_WEAK_UPDATE = 7,          /* Black/RedPixelUpdateWeak    APD.cu:4487-4489 */
	DVP_ST_GET_DEPTH_NORMAL = 8,     /* GetDepthandNormal      APD.cu:4494 */
	DVP_ST_FILTER_STRONG = 9,        /* Black/RedPixelFilterStrong  APD.cu:4497-4499 */
	DVP_ST_DEPTH_TO_WEAK = 10,       /* DepthToWeak            APD.cu:4502 */
	DVP_ST_LOCAL_REFINE = 11,        /* LocalRefine            APD.cu:4505 */
	DVP_ST_LAUNCHABLE = 12,          /* ids below this one are launch sites accepted by dvp_run_stage */
	DVP_ST_STRONG_PREP = 12,         /* timing bucket only: the pre-launch snapshot copies and the sample-search
	                                    launch issued by every DVP_ST_STRONG_UPDATE (DvpTimings) */
	DVP_ST_COUNT = 13
};

/* Device buffers (DataPassHelper members, APD.h:60-92) addressable by dvp_download_buffer /
 * dvp_upload_buffer. */
enum {
	DVP_BUF_PLANES = 0,              /* float4 per pixel   plane_hypotheses_cuda */
	DVP_BUF_COSTS = 1,               /* float              costs_cuda */
	DVP_BUF_SELECTED_VIEWS = 2,      /* uint32             selected_views_cuda */
	DVP_BUF_VIEW_WEIGHT = 3,         /* uint8 x 32         view_weight_cuda */
	DVP_BUF_WEAK_INFO = 4,           /* uint8              weak_info_cuda */
	DVP_BUF_WEAK_RELIABLE = 5,       /* uint8              weak_reliable_cuda */
	DVP_BUF_WEAK_NEAREST_STRONG = 6, /* short2             weak_nearest_strong */
	DVP_BUF_NEIGHBOURS_MAP = 7,      /* int32              neighbours_map_cuda */
	DVP_BUF_NEIGHBOURS = 8,          /* short2 x 12 per WEAK pixel   neighbours_cuda */
	DVP_BUF_FIT_PLANES = 9,          /* float4             fit_plane_hypotheses_cuda */
	DVP_BUF_CANDIDATE = 10,          /* short2 x 8 x (num_images-1) per pixel  candidate_cuda */
	DVP_BUF_EDGE = 11,               /* uint8              edge_cuda */
	DVP_BUF_EDGE_NEIGH = 12,         /* short2 x 8         edge_neigh_cuda */
	DVP_BUF_LABEL = 13,              /* int32              label_cuda */
	DVP_BUF_LABEL_BOUNDARY = 14,     /* short2 x 8 per WEAK pixel    label_boundary_cuda */
	DVP_BUF_COMPLEX = 15,            /* float per WEAK pixel         complex_cuda */
	DVP_BUF_RADIUS = 16,             /* int32              radius_cuda */
	DVP_BUF_COUNT = 17
};

/* Per-launch-site timing, measured with HIP events on the engine's own stream. */
typedef struct DvpTimings {
	double stage_ms[DVP_ST_COUNT];      /* accumulated kernel time per launch site */
	int32_t stage_launches[DVP_ST_COUNT];
	double iter_loop_ms;                /* the iteration loop, APD.cu:4478-4492 */
	double total_ms;                    /* whole dvp_run_patchmatch */
	uint64_t ncc_evals[DVP_ST_COUNT];   /* bilateral-NCC evaluations per launch site (only when
	                                       dvp_set_profiling(ctx, 1): counting build) */
} DvpTimings;

/* ---- lifetime (APD::APD / ~APD, APD.cpp:984-1043; cudaSetDevice, main.cpp:430-434) ---------- */
/* limits: 2 <= num_images <= 32 (APD.cpp:1083-1086), width, height <= 32767 (short2 pixel coordinates),
 * (width+4 rounded up to 64) * (height+4) * 8 bytes < 4 GiB (32-bit offsets into a row-pair plane) */
int dvp_ctx_create(int device, int width, int height, int num_images, dvp_ctx** out);
int dvp_ctx_destroy(dvp_ctx* ctx);
const char* dvp_last_error(const dvp_ctx* ctx);   /* ctx may be NULL: last create error */

/* ---- uploads (APD::CudaSpaceInitialization, APD.cpp:1497-1613) ------------------------------- */
/* images[i]: host pointer, row-major float32, `pitch_floats` elements per row (>= width);
 * replaces cudaMemcpy2DToArray + cudaCreateTextureObject (APD.cpp:1501-1517). */
int dvp_upload_images(dvp_ctx* ctx, const float* const* images, int pitch_floats);
/* depth maps of ref + src views for geom_consistency (APD.cpp:1522-1545) */
int dvp_upload_depths(dvp_ctx* ctx, const float* const* depths, int pitch_floats);
/* same, from device memory already resident on ctx's device (e.g. an RCCL broadcast buffer):
 * images/depths are copied device-to-device on the engine's stream. */
int dvp_upload_images_device(dvp_ctx* ctx, const float* const* dev_images, int pitch_floats);
int dvp_upload_depths_device(dvp_ctx* ctx, const float* const* dev_depths, int pitch_floats);
int dvp_upload_cameras(dvp_ctx* ctx, const DvpCamera* cams, int n);            /* APD.cpp:1549-1550 */
/* per-pixel input state; any pointer may be NULL (keeps the current/default content):
 * planes float4 (world normal, depth) APD.cpp:1566-1567 · selected_views APD.cpp:1560-1561 ·
 * weak_info APD.cpp:1595-1596 (also rebuilds neighbours_map/weak_count, APD.cpp:1182-1193) ·
 * edge APD.cpp:1577-1578 · label APD.cpp:1585-1586 · radius APD.cpp:1590-1591. */
int dvp_upload_state(dvp_ctx* ctx, const float* planes_xyzw, const uint32_t* selected_views,
                     const uint8_t* weak_info, const uint8_t* edge, const int32_t* label,
                     const int32_t* radius);
/* The same per-pixel input state for a REFINE_INIT pass whose previous maps come from the COARSER pyramid level
 * (src_w x src_h): replaces the five host-side RescaleMatToTargetSize calls of APD::InuputInitialization /
 * SupportInitialization (APD.cpp:1176-1180, 1440-1449, 1656-1659; the function itself APD.cpp:1773-1795, nearest
 * neighbour, row index / width ratio and column index / height ratio as the source has it), the plane assembly
 * (APD.cpp:1450-1456) and the radius rule for UNKNOWN pixels (APD.cpp:1660-1666) by one kernel on the maps at their own
 * size.  depth, normal_xyz (3 floats per pixel) and selected_views are required; weak_info NULL = every pixel STRONG
 * (APD.cpp:1196-1204); radius NULL = radius map untouched; edge / label are full-size maps as in dvp_upload_state. */
int dvp_upload_state_rescaled(dvp_ctx* ctx, int src_w, int src_h, const float* depth, const float* normal_xyz,
                              const uint32_t* selected_views, const uint8_t* weak_info, const int32_t* radius,
                              int radius_fallback, const uint8_t* edge, const int32_t* label);
/* device-side reset to the state a freshly constructed APD has before a FIRST_INIT pass without
 * prior: planes = 0 (out of range -> random init, APD.cu:1289-1291), selected_views = 0, every
 * pixel STRONG (APD.cpp:1196-1204), radius = strong_radius (APD.cpp:1649-1653), fit planes = 0
 * (APD.cpp:1571); edge/label maps are kept.  Lets one context process consecutive views without
 * host round trips. */
int dvp_reset_state(dvp_ctx* ctx);
/* device-side snapshot of the per-pixel input state (planes, selected_views, weak_info, radius) and
 * its restore (pass-internal buffers return to their freshly-uploaded content): re-running a pass
 * from identical inputs without re-uploading — what a fresh APD per view per pass (main.cpp:273,
 * APD.cpp:984-987) gives the reference for free. */
int dvp_save_state(dvp_ctx* ctx);
int dvp_restore_state(dvp_ctx* ctx);
/* Optional buffers of the context ahead of their first use — what a driver's helper thread calls on the context it prepares
 * for the next pyramid level, so that no multi-GB allocation lands inside a view's launches (the reference allocates everything in
 * CudaSpaceInitialization, APD.cpp:1497-1613).  flags: bit 0 = the split strong update's cost block, bit 1 = the view-compacted
 * DepthToWeak / LocalRefine passes' buffers; weak_pixels > 0: the weak update's anchor table and hand-over buffers for that many
 * WEAK pixels.  Never required: every launch site allocates what it lacks. */
int dvp_ctx_reserve(dvp_ctx* ctx, int weak_pixels, int flags);
int dvp_set_params(dvp_ctx* ctx, const DvpParams* params);                     /* APD.cpp:1607-1608 */
/* The reference seeds cuRAND with clock64() (APD.cu:1270); here the seed is explicit. */
int dvp_set_seed(dvp_ctx* ctx, uint64_t seed);
/* 0 = CUDA-texture-like 8-bit interpolation weights (default), 1 = exact fractions */
int dvp_set_sampler(dvp_ctx* ctx, int sampler);
int dvp_set_profiling(dvp_ctx* ctx, int count_evals);
/* Image planes the gather-bound kernels read after the last dvp_upload_images*: 0 = the float planes,
 * 1 = byte planes (kept besides the float planes when every texel of every image is an integer in
 * [0, 255]: images decoded from 8-bit files at their native size, APD.cpp:1057-1069).  Same values
 * either way, so results do not depend on it; DVP_NO_IMAGES8 in the environment forces 0. */
int dvp_image_format(const dvp_ctx* ctx);

/* ---- run (APD::RunPatchMatch, APD.cu:4406-4532) ---------------------------------------------- */
int dvp_run_patchmatch(dvp_ctx* ctx);
/* one launch site of the sequence; colour: 0 = Black*, 1 = Red* for the half launches (weak update also 2 = both colours as
 * one launch site, which is how dvp_run_patchmatch issues it: the two launches commute) */
int dvp_run_stage(dvp_ctx* ctx, int stage, int iter, int colour);
int dvp_synchronize(dvp_ctx* ctx);

/* ---- results (cudaMemcpy D2H, APD.cu:4525-4530; getters APD.cpp:1706-1748) ------------------- */
/* any pointer may be NULL. planes: (world normal xyz, depth w) per pixel. */
int dvp_download_state(dvp_ctx* ctx, float* planes_xyzw, uint32_t* selected_views,
                       uint8_t* weak_info, int32_t* radius);
/* The same results in the form the reference's driver stores them (ProcessProblem, main.cpp:300-309, which loops over
 * GetPlaneHypothesis): depth map = plane.w where depth_min <= w <= depth_max, else 0 and the pixel's state becomes
 * UNKNOWN (in the downloaded copy; the device state is untouched); normal map = 3 floats per pixel.  depth, normal_xyz
 * and weak_info are required. */
int dvp_download_maps(dvp_ctx* ctx, float* depth, float* normal_xyz, uint32_t* selected_views,
                      uint8_t* weak_info, int32_t* radius);
/* dvp_download_maps in two steps, for a driver that puts the next view on this context while the maps of the last one still
 * travel (the reference downloads synchronously, APD.cpp:1616-1640; a 25-Mpx view is 640 MB = 27 ms of PCIe time between 900 ms
 * of kernels).  _begin forms the maps in a staging buffer on the device — and, if depth_device_copy is not NULL, copies the
 * depth map to that DEVICE buffer (width * height floats: the resident map other views read as a source) — and returns when
 * the device is done with that; the context may then be reset, uploaded to and run again.  _finish copies the staged maps to
 * the host on a stream of its own and may be called from another thread; the next _begin (and dvp_ctx_destroy) waits for it. */
int dvp_download_maps_begin(dvp_ctx* ctx, float* depth_device_copy);
int dvp_download_maps_finish(dvp_ctx* ctx, float* depth, float* normal_xyz, uint32_t* selected_views,
                             uint8_t* weak_info, int32_t* radius);
long long dvp_buffer_bytes(dvp_ctx* ctx, int buffer);
int dvp_download_buffer(dvp_ctx* ctx, int buffer, void* dst);
int dvp_upload_buffer(dvp_ctx* ctx, int buffer, const void* src);
int dvp_weak_count(dvp_ctx* ctx);
int dvp_get_timings(dvp_ctx* ctx, DvpTimings* out);
int dvp_reset_timings(dvp_ctx* ctx);

/* ---- roofline micro-benchmark / known-answer tests ------------------------------------------- */
/* n (pixel, plane) pairs -> out[n * (num_images-1)] = ComputeMultiViewCostVectorOld
 * (APD.cu:1207-1216).  px = {x0,y0,x1,y1,...}; planes = camera-frame (nx,ny,nz,d) per pair.
 * Host pointers.  If kernel_ms != NULL it receives the kernel's HIP-event time. */
int dvp_eval_cost_vectors(dvp_ctx* ctx, const int32_t* px, const float* planes, int n, float* out,
                          float* kernel_ms);
/* device-resident variant for benchmarking: same computation on every pixel of the image with
 * plane = current plane_hypotheses (camera frame), `repeat` launches; returns the mean kernel ms */
/* sha256 of the kernel sources the library was built from (tools/csrc_hash.py), + the extra flags of a variant build */
const char* dvp_build_id(void);
int dvp_bench_cost_kernel(dvp_ctx* ctx, int repeat, float* mean_kernel_ms, uint64_t* evals_per_launch);

/* ---- depth-map fusion (RunFusion, APD.cpp:1809-1960) on the device ------------------------------
 * Replaces the body of the reference's host loop over views / pixels / sources: the geometric tests of every (pixel, source)
 * pair (Get3DPointonWorld + ProjectCamera + GetAngle, APD.cpp:1893-1931), the consistency vote (:1926-1929), the acceptance
 * (:1934) and the claims on the witnesses (`masks[...]`, :1888, 1911, 1942) — same points, same order, same bits as the
 * sequential scan (dvp-mvs_amd/csrc/dvp_fuse.hip says how the order-dependent claims are resolved in parallel).  The caller
 * keeps what the reference does around it: reading maps / images / cameras (:1836-1871) and writing the .ply (:1955-1958).
 * One job = one scene on one device.  acos / exp are the specified functions of csrc/dvp_fuse_math.hpp. */
typedef struct dvp_fuse dvp_fuse;
int dvp_fuse_create(int device, int num_views, dvp_fuse** out);
int dvp_fuse_destroy(dvp_fuse* job);
const char* dvp_fuse_last_error(const dvp_fuse* job);        /* job == NULL: the error of a failed dvp_fuse_create */
/* the maps of view slot `view` (host pointers, copied): camera with the intrinsics ALREADY rescaled to the maps' size
 * (RescaleImageAndCamera, APD.cpp:1750-1771), depth [rows*cols], normal [rows*cols*3], weak_info [rows*cols] or NULL
 * (every pixel STRONG), bgr [rows*cols*3] (the colour image at the maps' size), block [rows*cols] or NULL
 * (blocks/mask_<id>.jpg: reference pixels below 128 are skipped, APD.cpp:1885-1887) */
int dvp_fuse_set_view(dvp_fuse* job, int view, const DvpCamera* cam, int cols, int rows, const float* depth,
                      const float* normal_xyz, const uint8_t* weak_info, const uint8_t* bgr, const uint8_t* block);
/* one iteration of the reference's outer loop (APD.cpp:1874): view slot `view` scanned against the source slots `src`
 * (pair.txt order, sources without maps left out); accepted points are appended to the cloud in scan order */
int dvp_fuse_view(dvp_fuse* job, int view, const int* src, int num_src);
/* ... and of RunFusion_TAT_Intermediate (advanced = 0, APD.cpp:1962-2130) / RunFusion_TAT_advanced (advanced = 1,
 * APD.cpp:2132-2279): `src` holds ALL sources of the view in pair.txt order, -1 for a source without maps */
int dvp_fuse_view_graded(dvp_fuse* job, int view, const int* src, int num_src, int advanced);
long long dvp_fuse_count(const dvp_fuse* job);               /* points so far */
/* dvp_fuse_count() records of six floats — x y z b g r, struct PointList (main.h:69-72) — in scan order */
int dvp_fuse_download(dvp_fuse* job, float* points);
/* statistics of the last dvp_fuse_view: rounds of the parallel claim resolution, pixels left to the sequential finish */
int dvp_fuse_last_rounds(const dvp_fuse* job, int* rounds, int* rest);

#ifdef __cplusplus
}
#endif
#endif /* DVP_MVS_H_ */
