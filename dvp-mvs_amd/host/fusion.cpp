// fusion.cpp — depth-map fusion into a coloured point cloud (the reference's RunFusion, ETH3D variant,
// /root/reference/APD.cpp:1809-1960) and its binary PLY.
//
// What the step computes.  Views are visited in pair.txt order.  A pixel of view R with a positive depth
// that no earlier point has claimed is lifted to the world point X.  X is dropped into every source view
// S of R; the pixel it lands on (nearest pixel) is a WITNESS if it is unclaimed, has a depth, and agrees
// with R three ways: the witness' own world point re-projects within 2 px of the pixel in R, its depth
// seen from R is within 1 % of R's, and the two world normals are within 10 degrees.  Each witness votes
// exp(-(e + 200 r + 10 a)) (e: pixels, r: relative depth difference, a: radians).  The point is kept when
// it has a witness and the mean vote exceeds 0.3 (0.45 for a pixel R classified WEAK); it takes X as
// position, the mean colour of R's pixel and its witnesses, and claims the witnesses so that they do not
// produce a second copy when their own view's turn comes.  Claims make the result depend on the visiting
// order, which is therefore the reference's (view order, rows, columns, sources).
#include "APD.h"
#include "../csrc/dvp_fuse_math.hpp"   // acos / exp as specified functions: host, device kernels and the CPU restatement agree bit for bit
#include <cfloat>
#include <chrono>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace {

struct FusionView {
	int image_id = -1;
	Camera cam{};
	Mat depth, normal, weak, colour;   // CV_32FC1, CV_32FC3, CV_8UC1 (PixelState), CV_8UC3 (BGR) — all at the depth map's size
	Mat claimed;                       // CV_8UC1: pixel already merged into a point
	int cols() const { return depth.cols; }
	int rows() const { return depth.rows; }

	float centre[3] = { 0, 0, 0 };      // -R^T t in binary32: the fusion recomputes it instead of using Camera::c (APD.cpp:515-518)
	void set_centre() {
		for (int k = 0; k < 3; ++k) centre[k] = -(cam.R[0 + k] * cam.t[0] + cam.R[3 + k] * cam.t[1] + cam.R[6 + k] * cam.t[2]);
	}
	// pixel + depth -> world (camera frame through K^-1, then R^T . + C; APD.cpp:502-523)
	float3 lift(int x, int y, float z) const {
		const float cx = z * (x - cam.K[2]) / cam.K[0];
		const float cy = z * (y - cam.K[5]) / cam.K[4];
		float3 w;
		w.x = (cam.R[0] * cx + cam.R[3] * cy + cam.R[6] * z) + centre[0];
		w.y = (cam.R[1] * cx + cam.R[4] * cy + cam.R[7] * z) + centre[1];
		w.z = (cam.R[2] * cx + cam.R[5] * cy + cam.R[8] * z) + centre[2];
		return w;
	}
	const uint8_t* bgr(int x, int y) const { return colour.data + (size_t)y * colour.step + 3 * (size_t)x; }
};

struct Witness { int view, x, y; };

// angle between two unit normals; acos of a value rounded past 1 is NaN -> 0 (APD.cpp:1797-1806)
float normal_angle(const Vec3f& a, const Vec3f& b) {
	const float va[3] = { a[0], a[1], a[2] }, vb[3] = { b[0], b[1], b[2] };
	return dvp::fuse_angle(va, vb);
}

// colour image of a view brought to its depth map's resolution; the intrinsics follow the size
// (RescaleImageAndCamera, APD.cpp:1750-1771)
Mat fit_colour(const Mat& bgr, int cols, int rows, Camera* cam) {
	if (bgr.cols == cols && bgr.rows == rows) return bgr.clone();
	const float sx = cols / static_cast<float>(bgr.cols), sy = rows / static_cast<float>(bgr.rows);
	Mat out(rows, cols, CV_8UC3);
	Mat plane(bgr.rows, bgr.cols, CV_32FC1);
	for (int ch = 0; ch < 3; ++ch) {
		for (int y = 0; y < bgr.rows; ++y) {
			const uint8_t* s = bgr.data + (size_t)y * bgr.step + ch;
			float* d = plane.ptr<float>(y);
			for (int x = 0; x < bgr.cols; ++x) d[x] = s[3 * x];
		}
		const Mat small = ResizeLinear(plane, cols, rows);
		for (int y = 0; y < rows; ++y) {
			uint8_t* d = out.data + (size_t)y * out.step + ch;
			const float* s = small.ptr<float>(y);
			for (int x = 0; x < cols; ++x) d[3 * x] = (uint8_t)std::lrintf(std::min(255.f, std::max(0.f, s[x])));
		}
	}
	cam->K[0] *= sx; cam->K[2] *= sx;
	cam->K[4] *= sy; cam->K[5] *= sy;
	cam->width = cols; cam->height = rows;
	return out;
}

// blocks/mask_<id>.jpg: reference pixels below 128 are excluded whenever a blocks/ folder exists — in all three
// variants (APD.cpp:1831-1835 + 1885-1887, 1991-2013)
Mat load_block_mask(const path& dense_folder, int image_id) {
	return ReadImageGray(dense_folder / "blocks" / ("mask_" + std::to_string(image_id) + ".jpg"));
}

// The colour images of the views, decoded ahead of the fusion (PrefetchFusionImages: the driver starts it when the last pass begins —
// the host's cores idle while the GPU runs a full-size pass, and ten 25-Mpx colour JPEGs are 0.8 s of the 1.9 s the fusion took)
std::mutex g_colour_m;
std::map<int, std::shared_future<Mat>> g_colour_ahead;
Mat colour_image_of(const path& dense_folder, int image_id) {
	std::shared_future<Mat> f;
	{
		std::lock_guard<std::mutex> lk(g_colour_m);
		auto it = g_colour_ahead.find(image_id);
		if (it != g_colour_ahead.end()) { f = it->second; g_colour_ahead.erase(it); }
	}
	if (f.valid()) return f.get();
	return ReadImageColor(dense_folder / "images" / (ToFormatIndex(image_id) + ".jpg"));
}

bool load_view(const path& dense_folder, const Problem& problem, FusionView* v) {
	const std::string id = ToFormatIndex(problem.ref_image_id);
	v->image_id = problem.ref_image_id;
	ReadCameraOrDie(dense_folder / "cams" / (id + "_cam.txt"), v->cam);
	v->set_centre();
	if (!LoadResult(problem.result_folder / "depths.dmb", v->depth) || !LoadResult(problem.result_folder / "APD_normals.dmb", v->normal)) return false;
	Mat weak;
	LoadResult(problem.result_folder / "weak.bin", weak);
	if (weak.empty()) weak = Mat(v->rows(), v->cols(), CV_8UC1), std::memset(weak.data, STRONG, weak.step * weak.rows);
	v->weak = weak;
	RescaleMatToTargetSize<uint8_t>(weak, v->weak, v->cols(), v->rows());   // no-op when the sizes agree
	Mat bgr = colour_image_of(dense_folder, problem.ref_image_id);
	if (bgr.empty()) bgr = Mat::zeros(v->rows(), v->cols(), CV_8UC3);
	v->cam.width = bgr.cols;
	v->cam.height = bgr.rows;
	v->colour = fit_colour(bgr, v->cols(), v->rows(), &v->cam);
	v->claimed = Mat::zeros(v->rows(), v->cols(), CV_8UC1);
	return true;
}

}  // namespace

// ExportDepthImagePointCloud (APD.cpp:2281-2314; the reference's driver has the call commented out, main.cpp:413): every pixel of ONE
// depth map inside [depth_min, depth_max] lifted to the world, coloured from the view's image, columns outer / rows inner as the source
// has it, written as a .ply — the single-view debug cloud.
void ExportDepthImagePointCloud(const path& point_cloud_path, const path& image_path, const path& cam_path, Mat& depth, float depth_min, float depth_max) {
	FusionView v;
	ReadCameraOrDie(cam_path, v.cam);
	v.set_centre();
	Mat bgr = ReadImageColor(image_path);
	if (bgr.empty()) bgr = Mat::zeros(depth.rows, depth.cols, CV_8UC3);
	v.cam.width = bgr.cols;
	v.cam.height = bgr.rows;
	v.depth = depth;
	v.colour = fit_colour(bgr, depth.cols, depth.rows, &v.cam);   // RescaleImageAndCamera (APD.cpp:1750-1771)
	std::vector<PointList> cloud;
	for (int i = 0; i < depth.cols; i++)
		for (int j = 0; j < depth.rows; j++) {
			const float z = depth.at<float>(j, i);
			if (z < depth_min || z > depth_max || z != z) continue;
			const uint8_t* c = v.bgr(i, j);
			PointList pt;
			pt.coord = v.lift(i, j, z);
			pt.color = float3{ (float)c[0], (float)c[1], (float)c[2] };
			cloud.push_back(pt);
		}
	ExportPointCloud(point_cloud_path, cloud);
}

void PrefetchFusionImages(const path& dense_folder, const std::vector<Problem>& problems) {
	// a few helper threads, one image each at a time, in view order (the order the fusion asks for them)
	struct Queue { std::mutex m; size_t next = 0; std::vector<std::pair<int, std::shared_ptr<std::promise<Mat>>>> items; };
	auto q = std::make_shared<Queue>();
	{
		std::lock_guard<std::mutex> lk(g_colour_m);
		for (const Problem& p : problems) {
			if (g_colour_ahead.count(p.ref_image_id)) continue;
			auto pr = std::make_shared<std::promise<Mat>>();
			g_colour_ahead[p.ref_image_id] = pr->get_future().share();
			q->items.emplace_back(p.ref_image_id, pr);
		}
	}
	const int helpers = (int)std::min<size_t>(q->items.size(), (size_t)std::max(1, std::min(4, HostThreads() / 8)));
	for (int h = 0; h < helpers; ++h)
		std::thread([q, dense_folder]() {
			for (;;) {
				size_t i;
				{ std::lock_guard<std::mutex> lk(q->m); i = q->next++; }
				if (i >= q->items.size()) return;
				q->items[i].second->set_value(ReadImageColor(dense_folder / "images" / (ToFormatIndex(q->items[i].first) + ".jpg")));
			}
		}).detach();
}

namespace {
bool g_fusion_on_host = false;
int g_fusion_device = 0;
}
void SetFusionOnHost(bool on) { g_fusion_on_host = on; }
void SetFusionDevice(int device) { g_fusion_device = device; }

// RunFusion through the engine's C ABI (dvp_fuse_*, csrc/dvp_fuse.hip): this function keeps what the reference does around
// the scan — maps, colour images and cameras in (APD.cpp:1836-1871), the .ply out (:1955-1958)
// kind: 0 = RunFusion, 1 = RunFusion_TAT_Intermediate, 2 = RunFusion_TAT_advanced
static void RunFusionDevice(const path& dense_folder, const std::vector<Problem>& problems, int kind = 0) {
	const auto t_start = std::chrono::steady_clock::now();
	auto seconds_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
	const int n_views = (int)problems.size();
	std::vector<FusionView> views(n_views);
	std::vector<Mat> blocks(n_views);
	const bool use_block = std::filesystem::exists(dense_folder / "blocks");
	int max_id = -1;
	for (const Problem& p : problems) max_id = std::max(max_id, p.ref_image_id);
	std::vector<int> slot_of_id(max_id + 1, -1);
	for (int i = 0; i < n_views; ++i) {
		std::cout << "Reading image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		slot_of_id[problems[i].ref_image_id] = i;
	}
	dvp_fuse* job = nullptr;
	if (dvp_fuse_create(g_fusion_device, std::max(1, n_views), &job) != 0) DvpFatal(std::string("dvp_fuse_create failed: ") + dvp_fuse_last_error(nullptr));
	std::vector<char> loaded(n_views, 0);
	// maps from the result cache or the files + one colour JPEG per view on helper threads; the uploads in view order behind them
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(HostThreads(), 8))
	for (int i = 0; i < n_views; ++i) {
		loaded[i] = load_view(dense_folder, problems[i], &views[i]) ? 1 : 0;
		if (use_block) blocks[i] = load_block_mask(dense_folder, problems[i].ref_image_id);
		if (!loaded[i]) continue;
		FusionView& v = views[i];
		Mat bgr = v.colour;   // rows may carry padding: the ABI takes packed arrays
		std::vector<uint8_t> packed;
		const uint8_t* bgr_ptr = bgr.data;
		if (bgr.step != (size_t)3 * v.cols()) {
			packed.resize((size_t)3 * v.cols() * v.rows());
			for (int y = 0; y < v.rows(); ++y) std::memcpy(packed.data() + (size_t)3 * v.cols() * y, bgr.data + bgr.step * y, (size_t)3 * v.cols());
			bgr_ptr = packed.data();
		}
		const Mat& bl = blocks[i];
		const bool with_block = use_block && !bl.empty() && bl.cols == v.cols() && bl.rows == v.rows();
		int rc;
#pragma omp critical(dvp_fuse_upload)
		rc = dvp_fuse_set_view(job, i, &v.cam, v.cols(), v.rows(), v.depth.ptr<float>(0), v.normal.ptr<float>(0), v.weak.data, bgr_ptr, with_block ? bl.data : nullptr);
		if (rc != 0) DvpFatal(std::string("dvp_fuse_set_view failed: ") + dvp_fuse_last_error(job));
		if (use_block && !bl.empty() && !with_block) DvpFatal("RunFusion: a blocks/ mask of another size than the view's maps");
		v.depth = Mat(); v.normal = Mat(); v.colour = Mat(); v.weak = Mat();   // on the device now
	}
	for (int i = 0; i < n_views; ++i)
		if (!loaded[i]) std::cerr << "RunFusion: no depth/normal maps for view " << problems[i].ref_image_id << std::endl;
	const double t_load = seconds_since(t_start);
	const auto t_fuse0 = std::chrono::steady_clock::now();
	int rounds_max = 0;
	long long rest_sum = 0;
	for (int i = 0; i < n_views; ++i) {
		std::cout << "Fusing image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		if (!loaded[i]) continue;
		std::vector<int> sources;
		for (int id : problems[i].src_image_ids) {
			const int slot = (id >= 0 && id <= max_id && slot_of_id[id] >= 0 && loaded[slot_of_id[id]]) ? slot_of_id[id] : -1;
			if (kind != 0) sources.push_back(slot);          // the graded variants count every listed source (APD.cpp:2048)
			else if (slot >= 0) sources.push_back(slot);     // RunFusion skips the ones without maps
		}
		const int rc = kind == 0 ? dvp_fuse_view(job, i, sources.data(), (int)sources.size())
		                         : dvp_fuse_view_graded(job, i, sources.data(), (int)sources.size(), kind == 2 ? 1 : 0);
		if (rc != 0) DvpFatal(std::string("dvp_fuse_view failed: ") + dvp_fuse_last_error(job));
		int rounds = 0, rest = 0;
		dvp_fuse_last_rounds(job, &rounds, &rest);
		rounds_max = std::max(rounds_max, rounds);
		rest_sum += rest;
	}
	std::vector<PointList> cloud((size_t)dvp_fuse_count(job));
	static_assert(sizeof(PointList) == 24, "six floats");
	if (dvp_fuse_download(job, cloud.empty() ? nullptr : &cloud[0].coord.x) != 0) DvpFatal(std::string("dvp_fuse_download failed: ") + dvp_fuse_last_error(job));
	dvp_fuse_destroy(job);
	const double t_fuse = seconds_since(t_fuse0);
	const auto t_write0 = std::chrono::steady_clock::now();
	const path ply_path = dense_folder / "APD" / "APD.ply";
	ExportPointCloud(ply_path, cloud);
	std::cout << "Fusion: " << cloud.size() << " points -> " << ply_path << std::endl;
	std::cout << "  [fusion] device " << g_fusion_device << ": read maps + images + upload " << t_load << " s, fuse + download " << t_fuse << " s (at most " << rounds_max
	          << " resolve rounds per view, " << rest_sum << " pixels finished sequentially), write " << seconds_since(t_write0) << " s" << std::endl;
}

static void RunFusionHost(const path& dense_folder, const std::vector<Problem>& problems);
void RunFusion(const path& dense_folder, const std::vector<Problem>& problems) {
	if (g_fusion_on_host) RunFusionHost(dense_folder, problems);
	else RunFusionDevice(dense_folder, problems);
}

// The same scan on the host's cores (apd --fusion-on host; the form rounds 3-5 shipped): kept as a second implementation the
// tests compare with the device path and the sequential restatement.
static void RunFusionHost(const path& dense_folder, const std::vector<Problem>& problems) {
	const auto t_start = std::chrono::steady_clock::now();
	auto seconds_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
	const int n_views = (int)problems.size();
	std::vector<FusionView> views(n_views);
	std::vector<Mat> blocks(n_views);
	const bool use_block = std::filesystem::exists(dense_folder / "blocks");
	int max_id = -1;
	for (const Problem& p : problems) max_id = std::max(max_id, p.ref_image_id);
	std::vector<int> slot_of_id(max_id + 1, -1);   // image id -> position in `views`
	for (int i = 0; i < n_views; ++i) {
		std::cout << "Reading image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		slot_of_id[problems[i].ref_image_id] = i;
	}
	std::vector<char> loaded(n_views, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(HostThreads(), 8))   // (maps from the result cache or the files, one colour JPEG per view)
	for (int i = 0; i < n_views; ++i) {
		loaded[i] = load_view(dense_folder, problems[i], &views[i]) ? 1 : 0;
		if (use_block) blocks[i] = load_block_mask(dense_folder, problems[i].ref_image_id);
	}
	for (int i = 0; i < n_views; ++i)
		if (!loaded[i]) std::cerr << "RunFusion: no depth/normal maps for view " << problems[i].ref_image_id << std::endl;

	// The scan as two alternating steps over blocks of rows.  (1) For every pixel of the block, in parallel: the sources
	// whose pixel under X passes the three geometric tests, with their votes — nothing here depends on what has been claimed
	// so far (a claimed witness is only ever REMOVED from a pixel's list, and claims are never undone).  (2) In the
	// reference's order, on one thread: drop the witnesses claimed in the meantime, sum the votes of the others in source
	// order, accept, claim.  (2) is what makes the result order-dependent and it is a few loads and adds per pixel; (1) is
	// the two projections, the acos, the exp and the sqrt per (pixel, source) — 2.3 G of them for ten 25 Mpx views, two
	// minutes on one core, more than the whole PatchMatch schedule takes on the GPU.  Same points, same order, same bits
	// (tests/test_host_oracles.py compares with the sequential restatement).
	const double t_load = seconds_since(t_start);
	const auto t_fuse0 = std::chrono::steady_clock::now();
	double t_serial = 0.0;
	struct Candidate { int view, pixel; float vote; uint8_t bgr[3]; };   // pixel = y * cols + x in the source view; its colour
	struct Block { std::vector<Candidate> cand; std::vector<short> count; int y0 = 0, y1 = 0; };   // count: candidates of the pixel, -1 = not a reference pixel
	std::vector<PointList> cloud;
	{
		size_t pixels_with_depth = 0;
		for (const FusionView& v : views) pixels_with_depth += v.depth.empty() ? 0 : (size_t)v.rows() * v.cols();
		cloud.reserve(pixels_with_depth / 4);
	}
	std::vector<uint8_t*> flags_of(n_views, nullptr);   // claimed map of a view (CV_8UC1, rows contiguous: Mat::zeros)
	for (int v = 0; v < n_views; ++v) flags_of[v] = views[v].claimed.data;
	std::vector<uint8_t*> claim_list(4096);
	int n_claims = 0;
	Block buffers[2];
	for (int i = 0; i < n_views; ++i) {
		std::cout << "Fusing image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		FusionView& R = views[i];
		if (R.depth.empty()) continue;
		std::vector<int> sources;   // slots of this view's source images that are part of the job
		for (int id : problems[i].src_image_ids)
			if (id >= 0 && id <= max_id && slot_of_id[id] >= 0 && !views[slot_of_id[id]].depth.empty()) sources.push_back(slot_of_id[id]);
		const int NS = (int)sources.size();
		const int W = R.cols(), H = R.rows();
		const int block_rows = std::max(1, std::min(H, (int)((size_t)(64u << 20) / ((size_t)std::max(1, NS) * W * sizeof(Candidate)) + 1)));   // ~64 MB of candidates
		for (Block& bl : buffers) { bl.cand.resize((size_t)block_rows * W * std::max(1, NS)); bl.count.resize((size_t)block_rows * W); }
		// step 1 for the rows [y0, y1) into `bl`
		auto gather = [&](Block& bl, int y0, int y1) {
			bl.y0 = y0;
			bl.y1 = y1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(std::max(1, HostThreads() - 1))   // (one core stays with step 2)
			for (int y = y0; y < y1; ++y) {
				for (int x = 0; x < W; ++x) {
					const size_t p = (size_t)(y - y0) * W + x;
					bl.count[p] = -1;
					if (use_block && !blocks[i].empty() && blocks[i].at<uint8_t>(y, x) < 128) continue;
					const float z = R.depth.at<float>(y, x);
					if (R.claimed.at<uint8_t>(y, x) == 1 || z <= 0.0) continue;   // (R's own flags do not change during R's scan: claims go to witnesses, i.e. other views)
					const float3 X = R.lift(x, y, z);
					const Vec3f n_ref = R.normal.at<Vec3f>(y, x);
					Candidate* out = bl.cand.data() + p * NS;
					int n = 0;
					for (int s : sources) {
						const FusionView& S = views[s];
						float2 q;
						float zq;
						ProjectCamera(X, S.cam, q, zq);
						const int sx = int(q.x + 0.5f), sy = int(q.y + 0.5f);
						if (sx < 0 || sx >= S.cols() || sy < 0 || sy >= S.rows()) continue;
						const float zs = S.depth.at<float>(sy, sx);
						if (zs <= 0.0) continue;   // (claimed or not is step 2's question)
						float2 back;
						float z_seen;
						ProjectCamera(S.lift(sx, sy, zs), R.cam, back, z_seen);
						const float err = (float)std::sqrt(std::pow(x - back.x, 2) + std::pow(y - back.y, 2));
						const float rel = std::fabs(z_seen - z) / z;
						const float ang = normal_angle(n_ref, S.normal.at<Vec3f>(sy, sx));
						if (err < 2.0f && rel < 0.01f && ang < 0.174533f) {
							const uint8_t* c = S.bgr(sx, sy);
							out[n++] = Candidate{ s, sy * S.cols() + sx, dvp::fuse_expf(-(err + 200 * rel + ang * 10)), { c[0], c[1], c[2] } };
						}
					}
					bl.count[p] = (short)n;
				}
			}
		};
		// step 2 for the rows of `bl`, in the reference's order
		auto resolve = [&](const Block& bl) {
			const uint8_t* weak_rows = R.weak.data;
			for (int y = bl.y0; y < bl.y1; ++y) {
				const short* counts = bl.count.data() + (size_t)(y - bl.y0) * W;
				for (int x = 0; x < W; ++x) {
					const int nc = counts[x];
					if (nc <= 0) continue;   // not a reference pixel, or no witness
					const Candidate* in = bl.cand.data() + ((size_t)(y - bl.y0) * W + x) * NS;
					float votes = 0.0f, sb = 0.0f, sg = 0.0f, sr = 0.0f;
					int n = 0;
					uint64_t live = 0;   // bit k: candidate k is a witness (NS <= 64 sources; more fall back to the byte flags below)
					for (int k = 0; k < nc; ++k) {
						if (flags_of[in[k].view][in[k].pixel] == 1) continue;
						if (k < 64) live |= (uint64_t)1 << k;
						votes += in[k].vote;
						sb += in[k].bgr[0]; sg += in[k].bgr[1]; sr += in[k].bgr[2];
						++n;
					}
					const float needed = weak_rows[(size_t)y * R.weak.step + x] == WEAK ? 0.45f : 0.3f;
					if (n < 1 || !(votes > needed * n)) continue;
					for (int k = 0; k < nc; ++k)   // the witnesses are claimed (k >= 64: whatever was unclaimed a moment ago)
						if (k < 64 ? ((live >> k) & 1) != 0 : flags_of[in[k].view][in[k].pixel] != 1) claim_list[n_claims++] = flags_of[in[k].view] + in[k].pixel;
					for (int c = 0; c < n_claims; ++c) *claim_list[c] = 1;
					n_claims = 0;
					const uint8_t* c0 = R.bgr(x, y);
					// the reference adds the pixel's own colour first, then the witnesses' in source order (APD.cpp:1935-1946):
					// small integers in binary32, every partial sum exact, so the order of these additions cannot matter
					PointList pt;
					pt.coord = R.lift(x, y, R.depth.at<float>(y, x));
					pt.color = float3{ ((float)c0[0] + sb) / (n + 1), ((float)c0[1] + sg) / (n + 1), ((float)c0[2] + sr) / (n + 1) };
					cloud.push_back(pt);
				}
			}
		};
		// the blocks of the view: step 1 of block b + 1 runs (on the worker threads) while step 2 of block b runs here
		int cur = 0;
		gather(buffers[cur], 0, std::min(H, block_rows));
		for (int y0 = 0; y0 < H; y0 += block_rows) {
			const int n0 = y0 + block_rows, n1 = std::min(H, n0 + block_rows);
			std::future<void> next;
			if (n0 < H) next = std::async(std::launch::async, [&, n0, n1, cur] { gather(buffers[cur ^ 1], n0, n1); });
			const auto t_step2 = std::chrono::steady_clock::now();
			resolve(buffers[cur]);
			t_serial += seconds_since(t_step2);
			if (next.valid()) next.get();
			cur ^= 1;
		}
	}
	const double t_fuse = seconds_since(t_fuse0);
	const auto t_write0 = std::chrono::steady_clock::now();
	const path ply_path = dense_folder / "APD" / "APD.ply";
	ExportPointCloud(ply_path, cloud);
	std::cout << "Fusion: " << cloud.size() << " points -> " << ply_path << std::endl;
	std::cout << "  [fusion] read maps + images " << t_load << " s, fuse " << t_fuse << " s (of which the ordered step " << t_serial << " s), write " << seconds_since(t_write0) << " s, " << HostThreads() << " threads" << std::endl;
}

// ---- Tanks & Temples variants (RunFusion_TAT_Intermediate / _advanced, APD.cpp:1962-2279) ---------------------
// Same witness geometry as RunFusion, different acceptance: a pixel needs k mutually consistent witnesses
// for some k = 2..#sources, with thresholds that loosen with k (reprojection error < 0.25 k px, relative
// depth difference < k / 3500 resp. k / 3000; the intermediate variant also wants the normals within
// (4 + 3 k) degrees).  A kept pixel claims ITSELF (not its witnesses).  The intermediate variant averages
// the colours of the pixel and its witnesses, the advanced one keeps the pixel's colour.
// Reference quirk kept: the per-source residuals are one array reused across pixels and only overwritten
// when a source yields a comparison, so a source that drops out keeps voting with the residuals of the
// last pixel it was compared for (APD.cpp:2051, 2078-2090).
namespace {
void RunFusionGraded(const path& dense_folder, const std::vector<Problem>& problems, bool advanced) {
	const float dist_base = 0.25f, depth_base = advanced ? 1.0f / 3000.0f : 1.0f / 3500.0f;
	const float angle_base = 0.06981317007977318f, angle_grad = 0.05235987755982988f;   // 4 and 3 degrees
	const int n_views = (int)problems.size();
	std::vector<FusionView> views(n_views);
	std::vector<Mat> blocks(n_views);
	const bool use_block = std::filesystem::exists(dense_folder / "blocks");
	int max_id = -1;
	for (const Problem& p : problems) max_id = std::max(max_id, p.ref_image_id);
	std::vector<int> slot_of_id(max_id + 1, -1);
	for (int i = 0; i < n_views; ++i) {
		std::cout << "Reading image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		load_view(dense_folder, problems[i], &views[i]);
		if (use_block) blocks[i] = load_block_mask(dense_folder, problems[i].ref_image_id);
		slot_of_id[problems[i].ref_image_id] = i;
	}
	// Two steps per block of rows, as in RunFusion.  Here a pixel claims ITSELF and the claimed flags a pixel reads belong to
	// its source views, which do not change during this view's scan: the comparison of a (pixel, source) pair is a pure
	// function of the inputs (step 1, parallel).  What is ordered is the stale-residual quirk — `res[j]` keeps the values of
	// the last pixel source j was compared for — and the output order (step 2, one thread: copy the fresh residuals in,
	// the k = 2 .. n loop, emit).
	struct Residual { float err = FLT_MAX, rel = FLT_MAX, ang = FLT_MAX; int x = 0, y = 0; };
	struct Fresh { float err, rel, ang; int pixel; };   // pixel = y * cols + x in the source view, -1 = no comparison for this source
	struct Block { std::vector<Fresh> fresh; std::vector<uint8_t> is_ref; int y0 = 0, y1 = 0; };
	std::vector<PointList> cloud;
	{
		size_t pixels_with_depth = 0;
		for (const FusionView& v : views) pixels_with_depth += v.depth.empty() ? 0 : (size_t)v.rows() * v.cols();
		cloud.reserve(pixels_with_depth / 4);
	}
	Block buffers[2];
	for (int i = 0; i < n_views; ++i) {
		std::cout << "Fusing image " << std::setw(8) << std::setfill('0') << i << "..." << std::endl;
		FusionView& R = views[i];
		if (R.depth.empty()) continue;
		const std::vector<int>& src_ids = problems[i].src_image_ids;
		const int n_src = (int)src_ids.size();
		std::vector<int> slot(n_src, -1);   // view slot of source j, -1 = not part of the job / no maps
		for (int j = 0; j < n_src; ++j) {
			const int sv = (src_ids[j] >= 0 && src_ids[j] <= max_id) ? slot_of_id[src_ids[j]] : -1;
			slot[j] = (sv >= 0 && !views[sv].depth.empty()) ? sv : -1;
		}
		std::vector<Residual> res(n_src);   // NOT reset per pixel (see above)
		std::vector<char> agrees(n_src);
		const int W = R.cols(), H = R.rows();
		const int block_rows = std::max(1, std::min(H, (int)((size_t)(64u << 20) / ((size_t)std::max(1, n_src) * W * sizeof(Fresh)) + 1)));
		for (Block& bl : buffers) { bl.fresh.resize((size_t)block_rows * W * std::max(1, n_src)); bl.is_ref.resize((size_t)block_rows * W); }
		auto gather = [&](Block& bl, int y0, int y1) {
			bl.y0 = y0;
			bl.y1 = y1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(std::max(1, HostThreads() - 1))   // (one core stays with step 2)
			for (int y = y0; y < y1; ++y) {
				for (int x = 0; x < W; ++x) {
					const size_t p = (size_t)(y - y0) * W + x;
					bl.is_ref[p] = 0;
					if (use_block && !blocks[i].empty() && blocks[i].at<uint8_t>(y, x) < 128) continue;
					const float z = R.depth.at<float>(y, x);
					if (z <= 0.0) continue;
					bl.is_ref[p] = 1;
					const float3 X = R.lift(x, y, z);
					const Vec3f n_ref = R.normal.at<Vec3f>(y, x);
					Fresh* out = bl.fresh.data() + p * n_src;
					for (int j = 0; j < n_src; ++j) {
						out[j].pixel = -1;
						if (slot[j] < 0) continue;
						const FusionView& S = views[slot[j]];
						float2 q;
						float zq;
						ProjectCamera(X, S.cam, q, zq);
						const int sx = int(q.x + 0.5f), sy = int(q.y + 0.5f);
						if (sx < 0 || sx >= S.cols() || sy < 0 || sy >= S.rows()) continue;
						const float zs = S.depth.at<float>(sy, sx);
						if (S.claimed.at<uint8_t>(sy, sx) == 1 || zs <= 0.0) continue;   // (S's flags are final or still untouched: S is not the view being scanned)
						float2 back;
						float z_seen;
						ProjectCamera(S.lift(sx, sy, zs), R.cam, back, z_seen);
						out[j].err = (float)std::sqrt(std::pow(x - back.x, 2) + std::pow(y - back.y, 2));
						out[j].rel = std::fabs(z_seen - z) / z;
						out[j].ang = normal_angle(n_ref, S.normal.at<Vec3f>(sy, sx));
						out[j].pixel = sy * S.cols() + sx;
					}
				}
			}
		};
		auto resolve = [&](const Block& bl) {
			for (int y = bl.y0; y < bl.y1; ++y) {
				for (int x = 0; x < W; ++x) {
					const size_t p = (size_t)(y - bl.y0) * W + x;
					if (!bl.is_ref[p]) continue;
					const Fresh* in = bl.fresh.data() + p * n_src;
					for (int j = 0; j < n_src; ++j) {
						if (in[j].pixel < 0) continue;
						const int sc = views[slot[j]].cols();
						res[j].err = in[j].err; res[j].rel = in[j].rel; res[j].ang = in[j].ang;
						res[j].x = in[j].pixel % sc;
						res[j].y = in[j].pixel / sc;
					}
					for (int k = 2; k <= n_src; ++k) {
						int count = 0;
						for (int j = 0; j < n_src; ++j) {
							agrees[j] = res[j].err < k * dist_base && res[j].rel < k * depth_base && (advanced || res[j].ang < (k * angle_grad + angle_base));
							count += agrees[j];
						}
						if (count < k) continue;
						const uint8_t* c0 = R.bgr(x, y);
						float sum[3] = { (float)c0[0], (float)c0[1], (float)c0[2] };
						if (!advanced) {
							for (int j = 0; j < n_src; ++j) {
								if (!agrees[j]) continue;
								const FusionView& S = views[slot_of_id[src_ids[j]]];
								const uint8_t* cw = S.bgr(std::min(res[j].x, S.cols() - 1), std::min(res[j].y, S.rows() - 1));
								sum[0] += cw[0]; sum[1] += cw[1]; sum[2] += cw[2];
							}
							sum[0] /= (count + 1.0f); sum[1] /= (count + 1.0f); sum[2] /= (count + 1.0f);
						}
						PointList pt;
						pt.coord = R.lift(x, y, R.depth.at<float>(y, x));
						pt.color = float3{ sum[0], sum[1], sum[2] };
						cloud.push_back(pt);
						R.claimed.at<uint8_t>(y, x) = 1;
						break;
					}
				}
			}
		};
		int cur = 0;
		gather(buffers[cur], 0, std::min(H, block_rows));
		for (int y0 = 0; y0 < H; y0 += block_rows) {
			const int n0 = y0 + block_rows, n1 = std::min(H, n0 + block_rows);
			std::future<void> next;
			if (n0 < H) next = std::async(std::launch::async, [&, n0, n1, cur] { gather(buffers[cur ^ 1], n0, n1); });
			resolve(buffers[cur]);
			if (next.valid()) next.get();
			cur ^= 1;
		}
	}
	const path ply_path = dense_folder / "APD" / "APD.ply";
	ExportPointCloud(ply_path, cloud);
	std::cout << "Fusion: " << cloud.size() << " points -> " << ply_path << std::endl;
}
}  // namespace
void RunFusion_TAT_Intermediate(const path& dense_folder, const std::vector<Problem>& problems) {
	if (g_fusion_on_host) RunFusionGraded(dense_folder, problems, false);
	else RunFusionDevice(dense_folder, problems, 1);
}
void RunFusion_TAT_advanced(const path& dense_folder, const std::vector<Problem>& problems) {
	if (g_fusion_on_host) RunFusionGraded(dense_folder, problems, true);
	else RunFusionDevice(dense_folder, problems, 2);
}
