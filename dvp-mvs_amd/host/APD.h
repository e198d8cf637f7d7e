// APD.h — C++ host mirror of the reference's per-view pipeline API (/root/reference/APD.h:11-54,
// 94-199): same class name, method names (including the `InuputInitialization` spelling), call
// order and free functions, with the CUDA runtime replaced by the engine's C ABI
// (include/dvp_mvs.h), cv::Mat by Mat and boost::filesystem by std::filesystem.
// A `main.cpp`-style caller (ProcessProblem, /root/reference/main.cpp:267-419) compiles against
// this header unchanged apart from the include set.
#ifndef _APD_H_
#define _APD_H_
#include <ostream>
#include "main.h"
#include <functional>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// ---- free functions (APD.h:11-54) ------------------------------------------------------------------
void Connect(const Mat& dstImage, Mat& label_mask, std::vector<int>& label_cnt);      // APD.cpp:233-346
void Label_Update(Mat& label_mask, std::vector<int>& label_cnt);                       // APD.cpp:192-230
bool ReadBinMat(const path& mat_path, Mat& mat);                                       // APD.cpp:548-573
bool WriteBinMat(const path& mat_path, const Mat& mat);                                // APD.cpp:630-649
int writeDepthDmb(const path& mat_path, const Mat& depth);                             // APD.cpp:575-600
int writeNormalDmb(const path& mat_path, const Mat& normal);                           // APD.cpp:603-628
bool ReadCamera(const path& cam_path, Camera& cam);                                    // APD.cpp:651-692
void ReadCameraOrDie(const path& cam_path, Camera& cam);                                // ReadCamera + DvpFatal when the file is unusable (no caller may go on with an uninitialised camera)
bool ExportPointCloud(const path& point_cloud_path, std::vector<PointList>& pointcloud);   // APD.cpp:842-882
std::string ToFormatIndex(int index);                                                  // APD.cpp:978-982
template <typename TYPE>
void RescaleMatToTargetSize(const Mat& src, Mat& dst, int target_width, int target_height);   // APD.cpp:1773-1795
void RunFusion(const path& dense_folder, const std::vector<Problem>& problems);        // APD.cpp:1809-1960 (on the device: dvp_fuse_*, include/dvp_mvs.h)
void PrefetchFusionImages(const path& dense_folder, const std::vector<Problem>& problems);   // decode the views' colour images ahead of RunFusion (helper threads)
void SetFusionOnHost(bool on);          // RunFusion on the host's cores instead (same points, same order, same bits)
void SetFusionDevice(int device);       // the GPU RunFusion uses (default 0)
void ExportDepthImagePointCloud(const path& point_cloud_path, const path& image_path, const path& cam_path, Mat& depth, float depth_min, float depth_max);   // APD.cpp:2281-2314
void RunFusion_TAT_Intermediate(const path& dense_folder, const std::vector<Problem>& problems);   // APD.cpp:1962-2130
void RunFusion_TAT_advanced(const path& dense_folder, const std::vector<Problem>& problems);       // APD.cpp:2132-2279
Mat EdgeSegment(const int scale, const Mat& srcImage, int mode = 0, bool useCanny = false);   // APD.cpp:348-499 (mode 0 + Canny only)
void GetProblemEdges(const Problem& problem);                                           // main.cpp:193-246 (edge part)
void GetProblemEdges(const Problem& problem, const std::vector<path>& outputs);
std::vector<path> ProblemEdgeOutputs(const Problem& problem);   // the files GetProblemEdges would still have to produce
// Depth-Anything plane prior of a FIRST_INIT pass (APD.cpp:1210-1424), host/prior.cpp
std::vector<Triangle> DelaunayTriangulation(int cols, int rows, const Rect boundRC, std::vector<float2> xy_temps, std::vector<float> rates);   // APD.cpp:51-80
double calculateZ(const double A[3], const double B[3], const double C[3], double X, double Y);   // APD.cpp:30-49
void ProjectCamera(const float3 PointX, const Camera camera, float2& point, float& depth);          // APD.cpp:536-546
bool MetricDepthFromPrior(Mat& dep, const std::vector<float2>& xy, const std::vector<float3>& xyz, const Camera& cam);   // APD.cpp:1221-1356
void PlanesFromDepth(const Mat& dep, const Camera& cam, float4* planes);                              // APD.cpp:1365-1422
bool BuildPlanePrior(const Problem& problem, const Camera& scaled_ref_camera, int width, int height, float4* planes);
// image I/O without OpenCV: images/<id>.jpg through the built-in baseline decoder (host/jpeg.cpp), else
// <id>.pgm|.ppm (binary P5/P6); returns an empty Mat if nothing readable is found.
Mat DecodeJpeg(const path& file, int channels);     // 1: luma plane (libjpeg JCS_GRAYSCALE), 3: BGR
// intermediate maps of LabelSegment for the stage-by-stage comparison with an independent reading (tests/test_host_oracles.py)
struct LabelStages { Mat quarter, texture, texture_lines, resized, cleaned; int weak_tex_num = 0; };
Mat LabelSegment(const int scale, const Mat& src_image, LabelStages* stages = nullptr);   // EdgeSegment mode 1 (APD.cpp:348-401, 437-499), host/labels.cpp
Mat ReadImageGray(const path& image_path_jpg);      // stands in for cv::imread(IMREAD_GRAYSCALE), APD.cpp:1057
bool ImageFileSize(const path& image_path_jpg, int* width, int* height);   // from the file header, without decoding
Mat ReadImageColor(const path& image_path_jpg);     // cv::imread(IMREAD_COLOR) (BGR), APD.cpp:1842
Mat ResizeLinear(const Mat& src_f32, int new_cols, int new_rows);   // cv::resize(INTER_LINEAR), APD.cpp:1129
// threads of the host-side pixel loops: min(32, hardware threads), DVP_HOST_THREADS overrides
int HostThreads();
// Where a view's progress lines go: std::cout, or — while several views of a pass are in flight, one per driver thread — the
// calling thread's own buffer, which the driver prints in one piece when the view is done (SetViewLog(nullptr) ends it).
std::ostream& ViewLog();
void SetViewLog(std::ostream* buffer);
void SetThisThreadHostThreads(int n);   // caps HostThreads() for the calling thread (helper threads beside the driver's main thread); 0: no cap
void SetHostThreadShare(int world);   // caps HostThreads() at cores / world (one rank per GPU shares the host with its peers)
// write-back cache of the per-view result files + background workers (host/store.cpp)
void SetResultCache(bool enabled, size_t limit_bytes = 0);   // default: on, 32 GiB
void PublishResult(const path& file, const Mat& m);          // keep in memory + write in the background (.part + rename); `m` must not be modified afterwards
void ExpectResult(const path& file);                         // a background job will publish this path: loads of it wait
bool LoadResult(const path& file, Mat& m, bool will_modify = false);   // memory first, then the file
bool ResultExists(const path& file);
void RunInBackground(std::function<void()> job);             // one worker, at most two jobs queued
void FlushResults(bool drop_cache = false);                  // join jobs and writes
void WaitBackgroundJobs();                                   // join the jobs only (their results are in the cache; the files may still be on their way)
void ShutdownResultStore();
// error convention of the reference (CudaSafeCall, APD.cpp:943-951): message on stderr + exit.  Every such exit of the
// host library goes through DvpFatal; a multi-rank driver installs a hook that turns it into an agreed abort (comm.h).
[[noreturn]] void DvpFatal(const std::string& message);
void DvpSetFatalHook(void (*hook)(const char* message));
void DvpSafeCall(int rc, dvp_ctx* ctx, const char* what, const char* file, int line);
#define DVP_SAFE_CALL(ctx, expr) DvpSafeCall((expr), (ctx), #expr, __FILE__, __LINE__)

class APD {
public:
	APD(const Problem& problem);
	~APD();

	void InuputInitialization();
	void CudaSpaceInitialization();     // name kept; allocates/uploads through the HIP engine
	void SupportInitialization();
	void SetDataPassHelperInCuda();
	void RunPatchMatch();
	void RunPatchMatchToMaps(Mat& depth, Mat& normal);   // extension: results as the driver's depth / normal maps (APD.cpp)
	// ... in two steps (dvp_download_maps_begin / _finish): after StageMaps the context may go to the next view; the returned
	// function copies the maps to the host — depth and normal (allocated here, filled by the call) and the state maps
	// GetPixelStates / GetSelectedViews / GetRadiusMap returned — and may run on another thread after this object is gone.
	// depth_device_copy: a device buffer of width * height floats that receives the depth map (or nullptr).
	std::function<void()> RunPatchMatchAndStageMaps(Mat& depth, Mat& normal, float* depth_device_copy);
	float4 GetPlaneHypothesis(int r, int c);
	int GetPixelSelectedViews(int r, int c);
	void SetPixelSelectedViews(int r, int c, int temp_selected_views);
	Mat GetEdge();
	Mat GetPixelStates();
	Mat GetSelectedViews();
	Mat GetRadiusMap();
	int GetWidth();
	int GetHeight();
	float GetDepthMin();
	float GetDepthMax();
	// extensions (not in the reference): device selection, explicit seed, in-memory inputs
	static void SetDevice(int device);
	static void SetSeed(uint64_t seed);
	static void ReleasePooledContext();   // frees the recycled engine context and the image cache
	// The shipped reference writes labels_<s>.dmb but never loads it (the load is commented out,
	// APD.cpp:1630-1633): its label map stays zero unless MVS4/<id>.dmb needs rescaling.  true: load
	// labels_<s>.dmb in SupportInitialization (what the commented-out lines do).  Default false.
	static void SetUseLabelFiles(bool on);
	// true (default): a pass that starts from maps of another size (REFINE_INIT on a finer pyramid level) hands them to the
	// engine at their own size and RescaleMatToTargetSize runs there (dvp_upload_state_rescaled); false: the five host-side
	// rescales + the plane assembly of the reference's flow (APD.cpp:1176-1180, 1440-1456, 1656-1659).  Same maps either way.
	static void SetDeviceRescale(bool on);
	// multi-GPU hooks of the driver (comm.h).  Image cache: the decoded + rescaled float image of a view in
	// its reference role at the problem's scale — rank 0 fills it from disk, the others from a broadcast.
	static Mat CachedImage(const Problem& problem, int image_id, int* orig_cols, int* orig_rows);
	static void InsertCachedImage(const Problem& problem, int image_id, const Mat& image, int orig_cols, int orig_rows);
	static Mat DecodedGray(const path& image_file);   // cv::imread(GRAYSCALE) through a per-file cache (read-only result)
	static void PrefetchDecoded(const std::vector<path>& image_files);   // decode in the background (detached worker threads)
	static void PrewarmContext(int width, int height, int num_images);   // the next level's engine context, made by a helper thread
	static bool LevelSize(const Problem& problem, int scale, int* width, int* height);
	static void PrefetchLevelImages(std::vector<Problem> views, int scale);   // float images of a level ahead of its first pass
	static void ReserveImageCache(size_t views);   // the cache holds at least this many images before it evicts
	// Depth maps of the previous pass resident on this process' device (row-major, pitch = width).  When
	// the maps of a view and all its sources are registered, a geometric-consistency pass takes them from
	// there (dvp_upload_depths_device) instead of reading APD/<id>/depths.dmb (APD.cpp:1147-1166).
	// The float image of a view at the current pyramid level, resident on this process' device (row-major, pitch = width):
	// when the reference image and all sources of a view are registered at the view's size, CudaSpaceInitialization hands
	// them over with dvp_upload_images_device instead of re-sending ~100 MB per image and per reference view that uses it.
	static void SetResidentImage(int image_id, int scale, const float* device_ptr, int width, int height, int orig_width, int orig_height);
	static void ClearResidentImages();
	static void SetResidentDepth(int image_id, const float* device_ptr, int width, int height);
	static constexpr int kMaxViewsInFlight = 4;    // = the engine-context pool's slots (APD.cpp): more views in flight would recreate contexts per view
	static void UnsetResidentDepth(int image_id);   // before the registered block is freed / replaced
	static void ClearResidentDepths();
	static void SetResidentDownloader(void (*copy)(float* host, const float* device, size_t count));   // device -> host copy used when a resident map of another size has to be rescaled on the host
	const DvpTimings& GetTimings() const { return timings; }

private:
	int num_images = 0;
	int width = 0, height = 0;
	int ref_orig_width = 0, ref_orig_height = 0;   // the reference image before scaling (what sources are padded / cropped to)
	Problem problem;
	std::vector<Mat> images;
	std::vector<Mat> depths;
	std::vector<const float*> depths_device;   // non-empty: the pass uses resident depth maps
	std::vector<Camera> cameras;
	int weak_count = 0;
	Mat weak_info_host;
	float4* plane_hypotheses_host = nullptr;
	Mat edge_host, radius_host, label_host, selected_views_host;
	PatchMatchParams params_host;
	// device rescale: the previous pass' maps at the coarser level's size, handed to the engine by CudaSpaceInitialization
	bool coarse_state = false;
	Mat coarse_depth, coarse_normal, coarse_views, coarse_weak, coarse_radius;
	void CountWeak();
	void CoarseStateToHost();   // undo: rescale on the host after all (a map is missing or sizes disagree)
	dvp_ctx* ctx = nullptr;
	int ctx_device = 0;
	DvpTimings timings{};
};
#endif
