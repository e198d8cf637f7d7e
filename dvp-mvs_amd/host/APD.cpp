// APD.cpp — `class APD` on top of the HIP engine's C ABI.  Mirrors /root/reference/APD.cpp:984-1748:
// same methods, same call order (InuputInitialization -> SupportInitialization ->
// CudaSpaceInitialization -> SetDataPassHelperInCuda -> RunPatchMatch, main.cpp:276-280), same
// files read per pass.
#include "APD.h"
#include <mutex>
#include <chrono>
#include <memory>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <set>
#include <cstdlib>
#include <map>
#include <tuple>

static int g_device = 0;
static uint64_t g_seed = 0x5eed5eedULL;
void APD::SetDevice(int device) { g_device = device; }   // cudaSetDevice(argv[2]), main.cpp:430-434
void APD::SetSeed(uint64_t seed) { g_seed = seed; }      // the reference seeds with clock64() (APD.cu:1270)
static bool g_use_label_files = false;
void APD::SetUseLabelFiles(bool on) { g_use_label_files = on; }
// DVP_HOST_TIMING=1: wall time of the parts of the host steps (tools/e2e_timing.sh folds them per pass)
namespace {
struct HostLap {
	std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
	const bool on = std::getenv("DVP_HOST_TIMING") != nullptr;
	void operator()(const char* what) {
		if (!on) return;
		const auto now = std::chrono::steady_clock::now();
		ViewLog() << "  [host]   . " << what << ": " << std::chrono::duration_cast<std::chrono::microseconds>(now - t).count() / 1000.0 << " ms" << std::endl;
		t = now;
	}
};
}
static bool g_device_rescale = true;
void APD::SetDeviceRescale(bool on) { g_device_rescale = on; }

APD::APD(const Problem& problem) {   // APD.cpp:984-987
	params_host = problem.params;
	this->problem = problem;
}

// The reference allocates and frees ~30 device buffers per view (APD.cpp:1497-1613, 989-1043).  A
// context whose shape (device, W, H, images) matches the next view's is recycled instead: one pooled
// context per process, reset on the device (dvp_reset_state) and re-filled by the uploads.
namespace {
struct ImageEntry { Mat image; int orig_cols = 0, orig_rows = 0; uint64_t used = 0; };
using ImageKey = std::tuple<std::string, int, int, int>;   // file, scale, pad width, pad height
std::map<ImageKey, ImageEntry> g_img_cache;
// decoded 8-bit images (one per file, all pyramid levels are made from it): a 25 Mpx JPEG takes ~0.3 s to decode and the
// schedule visits every image once per level
std::map<std::string, Mat> g_decoded;
size_t g_decoded_bytes = 0;
constexpr size_t kDecodedLimit = (size_t)8 << 30;   // decoded 8-bit images kept across pyramid levels
bool g_decoded_full = false;                        // an image has been refused: prefetching more would decode them twice
std::recursive_mutex g_img_cache_mutex;   // the driver's background worker (edge maps of the next view) shares the cache
size_t g_img_cache_capacity = 96;
uint64_t g_img_cache_clock = 0;
// Room for one more entry: images of other pyramid levels go first (a level never comes back), then the least
// recently used one — never the whole cache (a scene with more views than the capacity would otherwise re-read
// every file for every view).
size_t g_img_cache_byte_limit = [] {   // float images are ~100 MB each at full resolution: the cache is bounded by bytes as well as by entries
	const char* e = std::getenv("DVP_IMAGE_CACHE_GB");
	return (size_t)(e ? std::max(1, std::atoi(e)) : 48) << 30;
}();
size_t cache_bytes_locked() {
	size_t b = 0;
	for (const auto& kv : g_img_cache) b += kv.second.image.step * (size_t)kv.second.image.rows;
	return b;
}
void make_room(int scale, size_t incoming_bytes) {
	auto full = [&] { return g_img_cache.size() >= g_img_cache_capacity || (!g_img_cache.empty() && cache_bytes_locked() + incoming_bytes > g_img_cache_byte_limit); };
	if (!full()) return;
	for (auto it = g_img_cache.begin(); it != g_img_cache.end();)
		it = (std::get<1>(it->first) != scale) ? g_img_cache.erase(it) : std::next(it);
	while (full()) {
		auto lru = g_img_cache.begin();
		for (auto it = g_img_cache.begin(); it != g_img_cache.end(); ++it)
			if (it->second.used < lru->second.used) lru = it;
		g_img_cache.erase(lru);
	}
}
struct PooledCtx {
	dvp_ctx* ctx = nullptr;
	int device = 0, w = 0, h = 0, ni = 0;
};
// no destructors: at static-destruction time the HIP runtime may already be gone; the driver calls
// APD::ReleasePooledContext() before it returns.  Several slots: the driver keeps up to that many views of a pass in
// flight at the coarse levels, each on its own context (main.cpp).
constexpr size_t kPoolSlots = (size_t)APD::kMaxViewsInFlight;
std::vector<PooledCtx>& g_pool = *new std::vector<PooledCtx>;
std::mutex g_ctx_mutex;   // g_pool, g_prewarm, the resident-map registries
dvp_ctx* take_pooled(int device, int w, int h, int ni) {
	std::lock_guard<std::mutex> lock(g_ctx_mutex);
	for (size_t i = 0; i < g_pool.size(); ++i)
		if (g_pool[i].device == device && g_pool[i].w == w && g_pool[i].h == h && g_pool[i].ni == ni) {
			dvp_ctx* c = g_pool[i].ctx;
			g_pool.erase(g_pool.begin() + (long)i);
			return c;
		}
	return nullptr;
}
void give_pooled(dvp_ctx* ctx, int device, int w, int h, int ni) {
	std::vector<dvp_ctx*> drop;
	{
		std::lock_guard<std::mutex> lock(g_ctx_mutex);
		// contexts of another shape belong to a level that is over (or to a view of another size): they go first, at once.
		// (Round 6 tried keeping them while the pool has room — ADVICE r05: views of mixed sizes in flight recreate contexts —
		// and measured the opposite of a gain on the ten-view schedule: the old level's contexts were then freed LATER, while
		// another view was inside its launches, and every hipFree stalls that view's own allocations: first weak update of a
		// level 490 ms instead of 6, profiles/r06_ab_notes.txt.)
		for (size_t i = 0; i < g_pool.size();)
			if (g_pool[i].device != device || g_pool[i].w != w || g_pool[i].h != h || g_pool[i].ni != ni) { drop.push_back(g_pool[i].ctx); g_pool.erase(g_pool.begin() + (long)i); }
			else ++i;
		if (g_pool.size() >= kPoolSlots) drop.push_back(ctx);
		else g_pool.push_back(PooledCtx{ ctx, device, w, h, ni });
	}
	for (dvp_ctx* c : drop) dvp_ctx_destroy(c);
}
}
static std::atomic<int> g_prefetch_threads{0};
// The next pyramid level's engine context, created by a helper thread while the GPU is still on the current level's last
// pass (allocating and clearing ~15 GB takes ~0.3 s at full resolution — once per level, but on the critical path of its
// first view otherwise).
namespace {
struct PrewarmedCtx {
	std::thread worker;
	dvp_ctx* ctx = nullptr;
	int device = 0, w = 0, h = 0, ni = 0;
	bool active = false;
	bool claimed = false;   // a caller of take_prewarmed is joining the worker
};
PrewarmedCtx& g_prewarm = *new PrewarmedCtx;   // never destroyed: an exit() while the helper runs must not meet a joinable std::thread's destructor
dvp_ctx* take_prewarmed(int device, int w, int h, int ni) {   // nullptr when there is none that fits
	std::thread worker;
	{
		std::lock_guard<std::mutex> lock(g_ctx_mutex);
		if (!g_prewarm.active || g_prewarm.claimed) return nullptr;
		g_prewarm.claimed = true;            // this caller joins the helper — outside the lock: the other driver threads
		worker = std::move(g_prewarm.worker);   // keep taking pooled contexts meanwhile
	}
	if (worker.joinable()) worker.join();
	dvp_ctx* c = nullptr;
	bool fits = false;
	{
		std::lock_guard<std::mutex> lock(g_ctx_mutex);
		c = g_prewarm.ctx;
		g_prewarm.ctx = nullptr;
		fits = g_prewarm.device == device && g_prewarm.w == w && g_prewarm.h == h && g_prewarm.ni == ni;
		g_prewarm.active = false;
		g_prewarm.claimed = false;
	}
	if (c && !fits) { dvp_ctx_destroy(c); c = nullptr; }
	return c;
}
}
void APD::PrewarmContext(int w, int h, int ni) {
	std::lock_guard<std::mutex> lock(g_ctx_mutex);
	if (g_prewarm.active || w <= 0 || h <= 0) return;
	g_prewarm.active = true;
	g_prewarm.device = g_device; g_prewarm.w = w; g_prewarm.h = h; g_prewarm.ni = ni;
	const int device = g_device;
	g_prewarm.worker = std::thread([device, w, h, ni]() {
		dvp_ctx* c = nullptr;
		if (dvp_ctx_create(device, w, h, ni, &c) != 0) c = nullptr;   // (the view that needs it will try again and report)
		// ... with its optional buffers: the next level's first views would otherwise allocate them inside their launches (the split
		// strong update's 7.8 GB, the sweep passes' 67 GB, the anchor table).  Fresh device memory costs 31-40 ms per GB on this part
		// (tools/micro/alloc_time.hip: hipMalloc of 16 / 48 / 120 GB = 0.54 / 1.9 / 3.7 s on an idle GPU; the views running meanwhile wait
		// for part of it), so the room for WEAK pixels (1408 S + 544 + 32 S bytes each: anchor table + hand-over records) is sized by
		// what the levels of a coarse-to-fine schedule hold, at most 90 % of the view: 48 GB below 8 Mpx — all of a 1552x1032 level
		// (its passes are 83-97 % WEAK), 55 % at 3104x2064 (17-38 %) — and 24 GB above — 7 % of a 25-Mpx view with 9 sources (1-6 %).
		// A view with more grows the table once, inside its weak update (1.1-1.7 s with the 16 GB of the first version at 3104x2064).
		if (c) {
			const long long per_px = 1408ll * (ni - 1) + 544 + 32ll * (ni - 1);
			const long long budget = (long long)w * h <= 8000000ll ? 48000000000ll : 24000000000ll;
			const long long room = std::min<long long>((long long)w * h * 9 / 10, budget / per_px);
			(void)dvp_ctx_reserve(c, (int)std::min<long long>(room, 2000000000ll), 3);
		}
		g_prewarm.ctx = c;
	});
}
// size of a view at a pyramid level, as load_image makes it (APD.cpp:1119-1131)
bool APD::LevelSize(const Problem& problem, int scale, int* w, int* h) {
	const Mat gray = DecodedGray(problem.dense_folder / path("images") / path(ToFormatIndex(problem.ref_image_id) + ".jpg"));
	if (gray.empty()) return false;
	const float factor = 1.0f / (float)scale;
	*w = scale != 1 ? (int)std::round(gray.cols * factor) : gray.cols;
	*h = scale != 1 ? (int)std::round(gray.rows * factor) : gray.rows;
	return true;
}
// the float images of these views at `scale` (reference role), made by a helper thread ahead of the level's first pass
void APD::PrefetchLevelImages(std::vector<Problem> views, int scale) {
	++g_prefetch_threads;
	std::thread([views, scale]() mutable {
		int lw = 0, lh = 0;
		const bool sized = !views.empty() && LevelSize(views[0], scale, &lw, &lh);
		for (Problem& p : views) {
			p.scale_size = scale;
			{   // never at the price of the level in use: a full cache evicts the OTHER levels' images first (make_room)
				std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex);
				if (g_img_cache.size() + 1 >= g_img_cache_capacity || (sized && cache_bytes_locked() + (size_t)lw * lh * 4 > g_img_cache_byte_limit)) break;
			}
			int oc, orr;
			(void)CachedImage(p, p.ref_image_id, &oc, &orr);
		}
		// ... and the level's result maps: every view of its first pass downloads into five maps nobody has released yet
		// (the result cache still holds the coarser level's) — fresh blocks whose pages would be faulted in by the
		// download.  Touch one set per view here and give it to the recycling pool (Mat.h).
		int w = 0, h = 0;
		if (!views.empty() && LevelSize(views[0], scale, &w, &h)) {
			std::vector<Mat> sets;
			size_t bytes = 0;
			const size_t keep = matpool::pool().cap;   // what the pool will not hold would be touched for nothing
			for (size_t v = 0; v < views.size() && bytes + 25 * (size_t)w * h <= keep; ++v)
				for (int type : { CV_32FC1, CV_32FC3, CV_32SC1, CV_32SC1, CV_8UC1 }) {
					sets.emplace_back(h, w, type);
					std::memset(sets.back().data, 0, sets.back().step * (size_t)h);
					bytes += sets.back().step * (size_t)h;
				}
		}
		--g_prefetch_threads;
	}).detach();
}
void APD::ReserveImageCache(size_t views) { std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex); g_img_cache_capacity = std::max<size_t>(96, 2 * views + 8); }   // reference + padded source role
void APD::ReleasePooledContext() {
	while (g_prefetch_threads.load() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));   // (they use the caches cleared below)
	{
		std::lock_guard<std::mutex> lock(g_ctx_mutex);
		for (PooledCtx& p : g_pool) dvp_ctx_destroy(p.ctx);
		g_pool.clear();
	}
	if (dvp_ctx* c = take_prewarmed(-1, 0, 0, 0)) dvp_ctx_destroy(c);   // (never fits: joined and freed)
	std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex);
	g_img_cache.clear();
	g_decoded.clear();
	g_decoded_bytes = 0;
	g_decoded_full = false;
}


// Decoded, padded and rescaled float images are cached per (file, scale, reference size): the
// reference re-reads and re-resizes every source image for every view that uses it
// (APD.cpp:1055-1143), 5-10x redundantly within a pass.  Same arithmetic, done once.
static ImageKey image_key(const Problem& problem, int image_id, int pad_w, int pad_h) {
	const path file = problem.dense_folder / path("images") / path(ToFormatIndex(image_id) + ".jpg");
	return ImageKey{ file.string(), problem.scale_size, pad_w, pad_h };
}
// decode outside the cache lock (a 25 Mpx JPEG takes ~0.3 s): a file being decoded by another thread is waited for
namespace {
std::set<std::string> g_decoding;
std::condition_variable_any g_decoded_cv;
}
Mat APD::DecodedGray(const path& file) {
	const std::string key = file.string();
	std::unique_lock<std::recursive_mutex> lock(g_img_cache_mutex);
	for (;;) {
		auto dit = g_decoded.find(key);
		if (dit != g_decoded.end()) return dit->second;
		if (!g_decoding.count(key)) break;
		g_decoded_cv.wait(lock);
	}
	g_decoding.insert(key);
	lock.unlock();
	Mat image_uint = ReadImageGray(file);
	lock.lock();
	g_decoding.erase(key);
	if (!image_uint.empty() && g_decoded_bytes + image_uint.step * image_uint.rows <= kDecodedLimit) {
		g_decoded[key] = image_uint;
		g_decoded_bytes += image_uint.step * image_uint.rows;
	} else if (!image_uint.empty()) g_decoded_full = true;
	g_decoded_cv.notify_all();
	return image_uint;
}
// all images of the job decoded on a few threads while the first passes run (the coarse levels are decode-bound otherwise)
void APD::PrefetchDecoded(const std::vector<path>& files) {
	const int n = std::max(1, std::min(HostThreads() / 2, 16));   // (ten 25-Mpx files on eight threads were two rounds of 0.26 s before the first kernel)
	// every file once (a source image appears in the list of every view that uses it), and nothing once the decoded cache
	// is full: an image that cannot be kept would be decoded here, dropped, and decoded again by load_image
	auto queue = std::make_shared<std::vector<path>>();
	std::set<std::string> listed;
	for (const path& f : files)
		if (listed.insert(f.string()).second) queue->push_back(f);
	auto next = std::make_shared<std::atomic<size_t>>(0);
	for (int t = 0; t < n; ++t) {
		++g_prefetch_threads;
		std::thread([queue, next]() {
			for (size_t i = (*next)++; i < queue->size(); i = (*next)++) {
				{
					std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex);
					if (g_decoded_full) break;
				}
				(void)APD::DecodedGray((*queue)[i]);
			}
			--g_prefetch_threads;
		}).detach();
	}
}
static ImageEntry load_image(const Problem& problem, int image_id, int pad_w, int pad_h, bool is_ref) {
	const ImageKey key = image_key(problem, image_id, pad_w, pad_h);
	{
		std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex);
		auto it = g_img_cache.find(key);
		if (it != g_img_cache.end()) { it->second.used = ++g_img_cache_clock; return it->second; }
		if (!is_ref) {   // the same file cached in its reference role needs no padding when the sizes agree
			auto ir = g_img_cache.find(image_key(problem, image_id, 0, 0));
			if (ir != g_img_cache.end() && ir->second.orig_cols == pad_w && ir->second.orig_rows == pad_h) { ir->second.used = ++g_img_cache_clock; return ir->second; }
		}
	}
	// decode / convert / resize WITHOUT the cache lock (the driver's background worker and the decode prefetch threads use
	// the same caches); two threads may build the same entry at the same time: same bits, the second insert is dropped
	const Mat image_uint = APD::DecodedGray(std::get<0>(key));
	if (image_uint.empty()) DvpFatal(std::string("Can't read ") + (is_ref ? "reference" : "source") + " image " + std::to_string(image_id));
	// uint8 -> float; a source image is zero-padded / cropped to the reference size (APD.cpp:1059, 1071-1079)
	const int fw = is_ref ? image_uint.cols : pad_w, fh = is_ref ? image_uint.rows : pad_h;
	Mat f = Mat::zeros(fh, fw, CV_32FC1);
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (int r = 0; r < std::min(fh, image_uint.rows); ++r) {
		const uint8_t* s = image_uint.ptr<uint8_t>(r);
		float* d = f.ptr<float>(r);
		for (int c = 0; c < std::min(fw, image_uint.cols); ++c) d[c] = (float)s[c];
	}
	ImageEntry ci;
	ci.orig_cols = f.cols;
	ci.orig_rows = f.rows;
	if (problem.scale_size != 1) {   // APD.cpp:1119-1131
		const float factor = 1.0f / (float)(problem.scale_size);
		f = ResizeLinear(f, (int)std::round(f.cols * factor), (int)std::round(f.rows * factor));
	}
	ci.image = f;
	std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex);
	make_room(problem.scale_size, f.step * (size_t)f.rows);
	ci.used = ++g_img_cache_clock;
	return g_img_cache.emplace(key, ci).first->second;
}
Mat APD::CachedImage(const Problem& problem, int image_id, int* orig_cols, int* orig_rows) {
	const ImageEntry ci = load_image(problem, image_id, 0, 0, true);
	*orig_cols = ci.orig_cols;
	*orig_rows = ci.orig_rows;
	return ci.image;
}
void APD::InsertCachedImage(const Problem& problem, int image_id, const Mat& image, int orig_cols, int orig_rows) {
	std::lock_guard<std::recursive_mutex> lock(g_img_cache_mutex);
	ImageEntry ci;
	ci.image = image;
	ci.orig_cols = orig_cols;
	ci.orig_rows = orig_rows;
	ci.used = ++g_img_cache_clock;
	make_room(problem.scale_size, image.step * (size_t)image.rows);
	g_img_cache[image_key(problem, image_id, 0, 0)] = ci;
}

namespace {
struct ResidentDepth { const float* ptr; int w, h; };
std::map<int, ResidentDepth> g_resident_depths;
std::mutex g_resident_mutex;   // both registries: the driver's view threads register and look up concurrently
}
void APD::SetResidentDepth(int image_id, const float* device_ptr, int width, int height) {
	std::lock_guard<std::mutex> lock(g_resident_mutex);
	g_resident_depths[image_id] = ResidentDepth{ device_ptr, width, height };
}
void APD::UnsetResidentDepth(int image_id) {
	std::lock_guard<std::mutex> lock(g_resident_mutex);
	g_resident_depths.erase(image_id);
}
void APD::ClearResidentDepths() {
	std::lock_guard<std::mutex> lock(g_resident_mutex);
	g_resident_depths.clear();
}
namespace {
struct ResidentImage { const float* ptr; int w, h, orig_w, orig_h; };
std::map<std::pair<int, int>, ResidentImage> g_resident_images;   // (image id, scale)
}
void APD::SetResidentImage(int image_id, int scale, const float* device_ptr, int width, int height, int orig_width, int orig_height) {
	std::lock_guard<std::mutex> lock(g_resident_mutex);
	g_resident_images[{ image_id, scale }] = ResidentImage{ device_ptr, width, height, orig_width, orig_height };
}
void APD::ClearResidentImages() {
	std::lock_guard<std::mutex> lock(g_resident_mutex);
	g_resident_images.clear();
}
namespace { void (*g_resident_download)(float*, const float*, size_t) = nullptr; }
void APD::SetResidentDownloader(void (*copy)(float*, const float*, size_t)) { g_resident_download = copy; }

APD::~APD() {                        // APD.cpp:989-1043
	delete[] plane_hypotheses_host;
	if (!ctx) return;
	give_pooled(ctx, ctx_device, width, height, num_images);
}

// APD.cpp:1045-1495 (the Depth-Anything prior block :1210-1424 lives in prior.cpp)
void APD::InuputInitialization() {
	HostLap lap;
	images.clear();
	cameras.clear();
	path image_folder = problem.dense_folder / path("images");
	path cam_folder = problem.dense_folder / path("cams");
	auto load = [&](int image_id, int pad_w, int pad_h, bool is_ref) -> ImageEntry {
		return load_image(problem, image_id, pad_w, pad_h, is_ref);
	};
	std::vector<std::pair<int, int>> orig_sizes;   // (cols, rows) before scaling, per image
	{
		const ImageEntry ci = load(problem.ref_image_id, 0, 0, true);
		images.push_back(ci.image);
		orig_sizes.emplace_back(ci.orig_cols, ci.orig_rows);
		width = ci.orig_cols;
		height = ci.orig_rows;
		ref_orig_width = ci.orig_cols;
		ref_orig_height = ci.orig_rows;
	}
	for (const auto& src_idx : problem.src_image_ids) {
		const ImageEntry ci = load(src_idx, width, height, false);
		images.push_back(ci.image);
		orig_sizes.emplace_back(ci.orig_cols, ci.orig_rows);
	}
	if (images.size() > MAX_IMAGES) {
		DvpFatal("Can't process so much images: " + std::to_string(images.size()));
	}
	{
		Camera cam;
		ReadCameraOrDie(cam_folder / path(ToFormatIndex(problem.ref_image_id) + "_cam.txt"), cam);
		cam.width = width;
		cam.height = height;
		cameras.push_back(cam);
	}
	for (const auto& src_idx : problem.src_image_ids) {
		Camera cam;
		ReadCameraOrDie(cam_folder / path(ToFormatIndex(src_idx) + "_cam.txt"), cam);
		cam.width = width;
		cam.height = height;
		cameras.push_back(cam);
	}
	params_host.depth_min = cameras[0].depth_min * 0.6f;   // APD.cpp:1109-1110
	params_host.depth_max = cameras[0].depth_max * 1.2f;
	params_host.num_images = (int)images.size();
	num_images = (int)images.size();
	ViewLog() << "Read images and camera done\n";
	ViewLog() << "Depth range: " << params_host.depth_min << " " << params_host.depth_max << std::endl;
	ViewLog() << "Num images: " << params_host.num_images << std::endl;
	if (problem.scale_size != 1) {   // APD.cpp:1119-1143 (the images were rescaled by load())
		for (int i = 0; i < num_images; ++i) {
			const int new_cols = images[i].cols, new_rows = images[i].rows;
			const float scale_x = new_cols / static_cast<float>(orig_sizes[i].first);
			const float scale_y = new_rows / static_cast<float>(orig_sizes[i].second);
			width = new_cols;
			height = new_rows;
			cameras[i].K[0] *= scale_x;
			cameras[i].K[2] *= scale_x;
			cameras[i].K[4] *= scale_y;
			cameras[i].K[5] *= scale_y;
			cameras[i].width = width;
			cameras[i].height = height;
		}
		ViewLog() << "Scale images and cameras done\n";
	}
	ViewLog() << "Image size: " << width << " * " << height << std::endl;
	lap("images + cameras");
	if (params_host.geom_consistency) {   // APD.cpp:1147-1166
		depths.clear();
		depths_device.clear();
		std::vector<int> ids(1, problem.ref_image_id);
		ids.insert(ids.end(), problem.src_image_ids.begin(), problem.src_image_ids.end());
		// a snapshot of the registry entries this view needs, taken under the registries' lock
		std::map<int, ResidentDepth> res_depths;
		bool resident;
		{
			std::lock_guard<std::mutex> lock(g_resident_mutex);
			resident = !g_resident_depths.empty();
			for (int id : ids) {
				auto it = g_resident_depths.find(id);
				if (it != g_resident_depths.end()) res_depths[id] = it->second;
				resident = resident && it != g_resident_depths.end() && it->second.w == width && it->second.h == height;
			}
		}
		if (resident) {   // previous-pass maps already on this device (multi-GPU exchange / --jacobi)
			for (int id : ids) depths_device.push_back(res_depths[id].ptr);
		} else {
			for (int id : ids) {
				Mat depth;
				// A map of another size (views of unequal size).  With a depth exchange (--jacobi / several ranks) the resident
				// copy IS the previous pass' map, on every rank alike: it is fetched and rescaled.  The owner's file would be the
				// previous pass' or this pass' map depending on how far the owner has come — never read it then.
				auto it = res_depths.find(id);
				if (it != res_depths.end() && g_resident_download) {
					depth = Mat(it->second.h, it->second.w, CV_32FC1);
					g_resident_download(depth.ptr<float>(0), it->second.ptr, (size_t)it->second.w * it->second.h);
				} else
				LoadResult(problem.dense_folder / path("APD") / path(ToFormatIndex(id)) / path("depths.dmb"), depth);
				if (depth.empty()) depth = Mat::zeros(height, width, CV_32FC1);
				if (depth.cols != width || depth.rows != height) RescaleMatToTargetSize<float>(depth, depth, width, height);
				depths.push_back(depth);
			}
		}
	}
	lap("source depth maps");
	if (params_host.use_APD) {            // APD.cpp:1169-1195
		path weak_info_path = problem.result_folder / path("weak.bin");
		if (!ResultExists(weak_info_path)) {
			DvpFatal("Can't find weak info file: " + weak_info_path.string());
		}
		LoadResult(weak_info_path, weak_info_host, true);   // RunPatchMatch downloads the new states into this buffer
		// With the device rescale the previous pass' maps stay as they are — of this pass' size, or of the coarser pyramid
		// level's in the first pass of a finer one — and the engine up-samples them, assembles the planes and applies the
		// radius rule (CudaSpaceInitialization; at equal size the rescale is the identity, as RescaleMatToTargetSize's early
		// return is); the depth / normal / selected-views maps below have to agree in size.
		coarse_state = g_device_rescale && params_host.state != FIRST_INIT && !weak_info_host.empty();
		if (coarse_state) {
			coarse_weak = weak_info_host;
			weak_info_host = Mat(height, width, CV_8UC1);
		} else {
			if (weak_info_host.cols != width || weak_info_host.rows != height) {
				std::cerr << "Weak info doesn't match the images' size!\n";
				RescaleMatToTargetSize<uint8_t>(weak_info_host, weak_info_host, width, height);
				ViewLog() << "Scale done\n";
			}
			CountWeak();
		}
	} else {                              // APD.cpp:1196-1204
		weak_info_host = Mat(height, width, CV_8UC1);
		weak_count = 0;
		std::memset(weak_info_host.data, STRONG, weak_info_host.step * (size_t)height);
	}
	lap("pixel states");
	plane_hypotheses_host = new float4[(size_t)width * height];
	if (params_host.state == FIRST_INIT) std::memset(plane_hypotheses_host, 0, sizeof(float4) * (size_t)width * height);   // (else every entry is assigned below)
	// FIRST_INIT: plane prior from the Depth-Anything map + sparse SfM points (dep/<id>.dmb,
	// sfm/<id>.txt, APD.cpp:1210-1424; host/prior.cpp).  Without those inputs the planes stay zero
	// (.w out of range) and RandomInitialization draws random planes (APD.cu:1289-1291).
	if (params_host.state == FIRST_INIT) {
		if (BuildPlanePrior(problem, cameras[0], width, height, plane_hypotheses_host)) ViewLog() << "Plane prior from dep/ and sfm/\n";
		else ViewLog() << "No dep/ + sfm/ prior: random plane initialisation\n";
	}
	if (params_host.state == FIRST_INIT) selected_views_host = Mat::zeros(height, width, CV_32SC1);   // (else loaded below)
	if (params_host.state != FIRST_INIT) {   // APD.cpp:1428-1456: the previous pass' maps are this pass' start
		Mat depth, normal;
		LoadResult(problem.result_folder / path("depths.dmb"), depth);
		LoadResult(problem.result_folder / path("APD_normals.dmb"), normal);
		LoadResult(problem.result_folder / path("selected_views.bin"), selected_views_host, true);   // downloaded into by RunPatchMatch
		if (depth.empty() || normal.empty() || selected_views_host.empty()) DvpFatal("Can't find the previous pass' maps in " + problem.result_folder.string());
		const bool fits = depth.cols == width && depth.rows == height && normal.cols == width && normal.rows == height;
		{   // device rescale: all maps of the previous pass at ONE size (else the host flow below)
			const bool same = normal.cols == depth.cols && normal.rows == depth.rows && selected_views_host.cols == depth.cols && selected_views_host.rows == depth.rows;
			const bool can = g_device_rescale && same && (params_host.use_APD ? (coarse_state && coarse_weak.cols == depth.cols && coarse_weak.rows == depth.rows) : true);
			if (coarse_state && !can) {   // the pixel states were kept back for nothing
				weak_info_host = coarse_weak;
				coarse_weak = Mat();
				coarse_state = false;
				RescaleMatToTargetSize<uint8_t>(weak_info_host, weak_info_host, width, height);
				CountWeak();
			}
			coarse_state = can;
		}
		if (coarse_state) {
			coarse_depth = depth;
			coarse_normal = normal;
			coarse_views = selected_views_host;
			selected_views_host = Mat(height, width, CV_32SC1);   // RunPatchMatch downloads into it
			lap("previous maps (kept at their size)");
			return;
		}
		if (!fits) {
			std::cerr << "Depth and Normal doesn't match the images' size!\n";
			RescaleMatToTargetSize<float>(depth, depth, width, height);
			RescaleMatToTargetSize<Vec3f>(normal, normal, width, height);
		}
		if (selected_views_host.cols != width || selected_views_host.rows != height) {
			std::cerr << "Select view doesn't match the images' size!\n";
			RescaleMatToTargetSize<unsigned int>(selected_views_host, selected_views_host, width, height);
		}
#pragma omp parallel for schedule(static) num_threads(HostThreads())
		for (int row = 0; row < height; ++row) {
			const float* z = depth.ptr<float>(row);
			const Vec3f* n = normal.ptr<Vec3f>(row);
			float4* out = plane_hypotheses_host + (size_t)row * width;
			for (int col = 0; col < width; ++col) out[col] = float4{ n[col][0], n[col][1], n[col][2], z[col] };
		}
		lap("previous maps -> planes");
	}
}

void APD::CountWeak() {   // APD.cpp:1182-1193 (the running index itself is made on the device, dvp_upload_state)
	long long wc = 0;
#pragma omp parallel for reduction(+ : wc) schedule(static) num_threads(HostThreads())
	for (int r = 0; r < height; ++r) {
		const uint8_t* row = weak_info_host.ptr<uint8_t>(r);
		for (int c = 0; c < width; ++c) wc += row[c] == WEAK;
	}
	weak_count = (int)wc;
	ViewLog() << "Weak count: " << weak_count << " / " << width * height << " = " << (float)weak_count / (float)(width * height) * 100 << "%" << std::endl;
}

// the host flow after all: what InuputInitialization would have done without the device rescale
void APD::CoarseStateToHost() {
	if (!coarse_state) return;
	coarse_state = false;
	// (RescaleMatToTargetSize returns early and leaves dst untouched when src already has the target size — every pass of a
	// level but its first: hand the map over as it is then)
	auto to_size = [&](auto tag, const Mat& src, Mat& dst) {
		using T = decltype(tag);
		if (src.cols == width && src.rows == height) dst = src;
		else RescaleMatToTargetSize<T>(src, dst, width, height);
	};
	if (!coarse_weak.empty()) {
		to_size(uint8_t(), coarse_weak, weak_info_host);
		CountWeak();
	}
	Mat depth, normal;
	to_size(float(), coarse_depth, depth);
	to_size(Vec3f(), coarse_normal, normal);
	to_size((unsigned int)0, coarse_views, selected_views_host);
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (int row = 0; row < height; ++row) {
		const float* z = depth.ptr<float>(row);
		const Vec3f* n = normal.ptr<Vec3f>(row);
		float4* out = plane_hypotheses_host + (size_t)row * width;
		for (int col = 0; col < width; ++col) out[col] = float4{ n[col][0], n[col][1], n[col][2], z[col] };
	}
	coarse_depth = coarse_normal = coarse_views = coarse_weak = coarse_radius = Mat();
}

// APD.cpp:1615-1668
void APD::SupportInitialization() {
	int scale = 0;
	while ((1 << scale) < problem.scale_size) scale++;
	if (problem.params.use_edge || problem.params.use_limit) {
		path edge_path = problem.result_folder / path("edges_" + std::to_string(scale) + ".dmb");
		if (!LoadResult(edge_path, edge_host) || edge_host.cols != width || edge_host.rows != height) {
			// edges_<s>.dmb is written by GetProblemEdges (edges.cpp) before the first pass; a caller
			// that skipped it gets an empty edge map (no edge pixels)
			edge_host = Mat::zeros(height, width, CV_8UC1);
		}
	}
	if (problem.params.use_label) {
		// the shipped reference only fills label_host when MVS4/<id>.dmb needs rescaling
		// (APD.cpp:1636-1645) and uploads it regardless; here: load if present, else zeros
		Mat ref_dep;
		path p = problem.dense_folder / path("MVS4") / path(ToFormatIndex(problem.ref_image_id) + ".dmb");
		label_host = Mat();   // empty = all zero: nothing is uploaded, the context's label map is zero (fresh, or dvp_reset_state)
		if (std::filesystem::exists(p) && ReadBinMat(p, ref_dep) && (ref_dep.cols != width || ref_dep.rows != height)) {
			Mat tmp;
			RescaleMatToTargetSize<float>(ref_dep, tmp, width, height);
			label_host = Mat(height, width, CV_32SC1);
			std::memcpy(label_host.data, tmp.data, (size_t)width * height * 4);   // float bits reinterpreted, as the reference does
		} else if (g_use_label_files) {   // the commented-out load of APD.cpp:1630-1633
			Mat lab;
			if (LoadResult(problem.result_folder / path("labels_" + std::to_string(scale) + ".dmb"), lab) && lab.type() == CV_32SC1) {
				if (lab.cols != width || lab.rows != height) RescaleMatToTargetSize<int>(lab, lab, width, height);
				label_host = lab;
			}
		}
	}
	if (problem.params.use_radius) {   // APD.cpp:1648-1667: the first pass starts at strong_radius, later ones at the stored map
		const int fallback = problem.params.strong_radius;
		if (problem.params.state != FIRST_INIT) LoadResult(problem.result_folder / path("radius.bin"), radius_host, true);
		if (problem.params.state == FIRST_INIT || radius_host.empty()) {
			radius_host = Mat(height, width, CV_32S);
			std::fill(radius_host.ptr<int>(0), radius_host.ptr<int>(0) + (size_t)width * height, fallback);
		}
		if (coarse_state) {   // the radius map travels with the other coarse maps; the UNKNOWN rule below runs on the device too
			if (radius_host.cols == coarse_depth.cols && radius_host.rows == coarse_depth.rows) {
				coarse_radius = radius_host;
				radius_host = Mat(height, width, CV_32S);   // RunPatchMatch downloads into it
				return;
			}
			CoarseStateToHost();   // no radius.bin of the coarse size (fallback map above): everything on the host
		}
		if (radius_host.cols != width || radius_host.rows != height) {
			std::cerr << "Radius map doesn't match the images' size!\n";
			RescaleMatToTargetSize<int>(radius_host, radius_host, width, height);
		}
		const uint8_t* state = weak_info_host.ptr<uint8_t>(0);
		int* rad = radius_host.ptr<int>(0);
		const long long npx = (long long)width * height;
#pragma omp parallel for schedule(static) num_threads(HostThreads())
		for (long long i = 0; i < npx; ++i)
			if (state[i] == UNKNOWN) rad[i] = fallback;   // a pixel that lost its estimate restarts with the default patch
	}
}

// APD.cpp:1497-1613: every cudaMalloc/cudaMemcpy/texture creation becomes one C-ABI upload
void APD::CudaSpaceInitialization() {
	HostLap lap;
	ctx_device = g_device;
	if ((ctx = take_pooled(g_device, width, height, num_images)) != nullptr) {
		DVP_SAFE_CALL(ctx, dvp_reset_state(ctx));
		DVP_SAFE_CALL(ctx, dvp_reset_timings(ctx));
	} else if ((ctx = take_prewarmed(g_device, width, height, num_images)) != nullptr) {
	} else if (dvp_ctx_create(g_device, width, height, num_images, &ctx) != 0) {
		DvpFatal(std::string("dvp_ctx_create failed: ") + dvp_last_error(nullptr));
	}
	lap("context");
	std::vector<const float*> ptrs(num_images);
	bool resident_images;
	{
	std::lock_guard<std::mutex> resident_lock(g_resident_mutex);   // held over the look-ups only
	resident_images = !g_resident_images.empty();
	for (int i = 0; i < num_images && resident_images; ++i) {
		// A source image of another ORIGINAL size than the reference is padded / cropped to the reference's original size on the
		// host and resized after that (APD.cpp:1071-1079): the resident copy — the file resized at its own size — is not that
		// image even when the rounded scaled sizes coincide (1001x800 and 1000x800 at scale 4 are both 250x200), so the
		// original sizes have to agree, as load_image's shortcut requires.
		auto it = g_resident_images.find({ i == 0 ? problem.ref_image_id : problem.src_image_ids[i - 1], problem.scale_size });
		resident_images = it != g_resident_images.end() && it->second.w == width && it->second.h == height &&
		                  it->second.orig_w == ref_orig_width && it->second.orig_h == ref_orig_height;
		if (resident_images) ptrs[i] = it->second.ptr;
	}
	}
	if (resident_images) DVP_SAFE_CALL(ctx, dvp_upload_images_device(ctx, ptrs.data(), width));
	else {
		for (int i = 0; i < num_images; ++i) ptrs[i] = images[i].ptr<float>(0);
		DVP_SAFE_CALL(ctx, dvp_upload_images(ctx, ptrs.data(), width));
	}
	lap("images upload");
	if (params_host.geom_consistency) {
		if (!depths_device.empty()) {
			DVP_SAFE_CALL(ctx, dvp_upload_depths_device(ctx, depths_device.data(), width));
		} else {
			for (int i = 0; i < num_images; ++i) ptrs[i] = depths[i].ptr<float>(0);
			DVP_SAFE_CALL(ctx, dvp_upload_depths(ctx, ptrs.data(), width));
		}
	}
	DVP_SAFE_CALL(ctx, dvp_upload_cameras(ctx, reinterpret_cast<const DvpCamera*>(cameras.data()), num_images));
	lap("depth maps + cameras upload");
	if (coarse_state) {   // REFINE_INIT from the coarser level: RescaleMatToTargetSize + plane assembly + radius rule on the device
		DVP_SAFE_CALL(ctx, dvp_upload_state_rescaled(ctx, coarse_depth.cols, coarse_depth.rows, coarse_depth.ptr<float>(0), coarse_normal.ptr<float>(0),
			coarse_views.ptr<uint32_t>(0), coarse_weak.empty() ? nullptr : coarse_weak.ptr<uint8_t>(0),
			coarse_radius.empty() ? nullptr : coarse_radius.ptr<int32_t>(0), problem.params.strong_radius,
			(problem.params.use_edge || problem.params.use_limit) ? edge_host.ptr<uint8_t>(0) : nullptr,
			(problem.params.use_label && !label_host.empty()) ? label_host.ptr<int32_t>(0) : nullptr));
		weak_count = dvp_weak_count(ctx);
		ViewLog() << "Weak count: " << weak_count << " / " << width * height << " = " << (float)weak_count / (float)(width * height) * 100 << "%" << std::endl;
		coarse_depth = coarse_normal = coarse_views = coarse_weak = coarse_radius = Mat();
		lap("state upload (coarse maps, rescaled on the device)");
		return;
	}
	DVP_SAFE_CALL(ctx, dvp_upload_state(ctx, reinterpret_cast<const float*>(plane_hypotheses_host),
		selected_views_host.ptr<uint32_t>(0), weak_info_host.ptr<uint8_t>(0),
		(problem.params.use_edge || problem.params.use_limit) ? edge_host.ptr<uint8_t>(0) : nullptr,
		(problem.params.use_label && !label_host.empty()) ? label_host.ptr<int32_t>(0) : nullptr,
		problem.params.use_radius ? radius_host.ptr<int32_t>(0) : nullptr));
	lap("state upload");
}

// APD.cpp:1670-1704: the DataPassHelper lives inside the context; what is left is params + seed
void APD::SetDataPassHelperInCuda() {
	DVP_SAFE_CALL(ctx, dvp_set_params(ctx, reinterpret_cast<const DvpParams*>(&params_host)));
	DVP_SAFE_CALL(ctx, dvp_set_seed(ctx, g_seed + (uint64_t)problem.ref_image_id * 1000003ull + (uint64_t)problem.iteration));
}

// APD.cu:4406-4532
void APD::RunPatchMatch() {
	DVP_SAFE_CALL(ctx, dvp_run_patchmatch(ctx));
	if (!problem.params.use_radius) radius_host = Mat::zeros(height, width, CV_32S);
	DVP_SAFE_CALL(ctx, dvp_download_state(ctx, reinterpret_cast<float*>(plane_hypotheses_host), selected_views_host.ptr<uint32_t>(0),
		weak_info_host.ptr<uint8_t>(0), problem.params.use_radius ? radius_host.ptr<int32_t>(0) : nullptr));
	DVP_SAFE_CALL(ctx, dvp_get_timings(ctx, &timings));
}

// extension: the same run, results downloaded as the maps the driver stores (main.cpp:300-309 done on the device,
// dvp_download_maps) instead of planes the caller unpacks; GetPlaneHypothesis is not served afterwards.
void APD::RunPatchMatchToMaps(Mat& depth, Mat& normal) {
	DVP_SAFE_CALL(ctx, dvp_run_patchmatch(ctx));
	if (!problem.params.use_radius) radius_host = Mat::zeros(height, width, CV_32S);
	depth = Mat(height, width, CV_32FC1);
	normal = Mat(height, width, CV_32FC3);
	DVP_SAFE_CALL(ctx, dvp_download_maps(ctx, depth.ptr<float>(0), normal.ptr<float>(0), selected_views_host.ptr<uint32_t>(0),
		weak_info_host.ptr<uint8_t>(0), problem.params.use_radius ? radius_host.ptr<int32_t>(0) : nullptr));
	DVP_SAFE_CALL(ctx, dvp_get_timings(ctx, &timings));
}

std::function<void()> APD::RunPatchMatchAndStageMaps(Mat& depth, Mat& normal, float* depth_device_copy) {
	DVP_SAFE_CALL(ctx, dvp_run_patchmatch(ctx));
	if (!problem.params.use_radius) radius_host = Mat::zeros(height, width, CV_32S);
	depth = Mat(height, width, CV_32FC1);
	normal = Mat(height, width, CV_32FC3);
	DVP_SAFE_CALL(ctx, dvp_download_maps_begin(ctx, depth_device_copy));
	DVP_SAFE_CALL(ctx, dvp_get_timings(ctx, &timings));
	// (the Mats share their buffers with the copies held here: they stay alive with the function)
	dvp_ctx* const c = ctx;
	Mat d = depth, n = normal, v = selected_views_host, w = weak_info_host, r = problem.params.use_radius ? radius_host : Mat();
	return [c, d, n, v, w, r]() mutable {
		if (dvp_download_maps_finish(c, d.ptr<float>(0), n.ptr<float>(0), v.ptr<uint32_t>(0), w.ptr<uint8_t>(0), r.empty() ? nullptr : r.ptr<int32_t>(0)) != 0)
			DvpFatal("dvp_download_maps_finish failed");   // (the context may be gone by now: its error text is not read)
	};
}

float4 APD::GetPlaneHypothesis(int r, int c) { return plane_hypotheses_host[c + r * width]; }   // APD.cpp:1706-1708
int APD::GetPixelSelectedViews(int r, int c) { return selected_views_host.at<int>(r, c); }
void APD::SetPixelSelectedViews(int r, int c, int v) { selected_views_host.at<int>(r, c) = v; }
Mat APD::GetEdge() { return edge_host; }
Mat APD::GetPixelStates() { return weak_info_host; }
Mat APD::GetSelectedViews() { return selected_views_host; }
Mat APD::GetRadiusMap() { return radius_host; }
int APD::GetWidth() { return width; }
int APD::GetHeight() { return height; }
float APD::GetDepthMin() { return params_host.depth_min; }
float APD::GetDepthMax() { return params_host.depth_max; }
