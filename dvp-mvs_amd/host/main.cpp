// main.cpp — the driver: the reference's coarse-to-fine schedule (/root/reference/main.cpp:421-528) over
// `class APD`, one process per GPU.
//   apd <dense_folder> [gpu_index] [--max-src N] [--iters N] [--min-scale S] [--passes P] [--seed X]
//       [--rank R --world N --job ID [--transport rccl|host] [--collective-timeout SEC]] [--jacobi] [--labels]
//       [--no-fusion | --fusion eth|tat-intermediate|tat-advanced] [--fusion-on device|host]
//
// Schedule.  The image pyramid has round_num levels (the longer side is halved until <= 800).  Level i
// runs one "A" pass without geometric consistency — FIRST_INIT from scratch / the Depth-Anything prior
// at the coarsest level, REFINE_INIT from the up-sampled result of the level below otherwise — and then
// `passes` REFINE_ITER passes with geometric consistency against the source views' depth maps.  The
// reference stops before the finest level (`i < round_num - 1`); --min-scale 1 adds it.
//
// Multi-GPU.  Views of a pass are independent: view v belongs to rank v % world.  Every rank decodes and
// resizes the images of ITS views at a level and broadcasts them (RCCL), so that the decode work scales with the
// rank count and every file is read once; after every pass each view's new depth map is
// broadcast by its owner and kept resident on every GPU, which is where the next pass' geometric term
// reads it (Jacobi: a pass sees the previous pass' maps only).  --job must be unique per run (default: $DVP_JOB_ID,
// $TORCHELASTIC_RUN_ID or $SLURM_JOB_ID); a rank that fails takes the whole job down (comm.h, "Failure model").  With one rank the default is the
// reference's in-place order (a view sees the maps its predecessors wrote in the same pass, through the
// result files); --jacobi selects the exchange semantics there too, which makes a run independent of the
// number of ranks.  Result files are always written to a temporary name and renamed, so a reader never
// sees a torn file.
#include <sstream>
#include "APD.h"
#include <thread>
#include "comm.h"
#include <cstdlib>
#include <map>
#include <set>
#include <memory>
#include <future>
#include <mutex>
#include <condition_variable>
#include <algorithm>

namespace {

struct Options {
	path dense_folder;
	int gpu = 0, max_src = 0, iters = 3, min_scale = 2, geom_passes = 3, rank = 0, world = 1;
	uint64_t seed = 1234;
	bool fusion = true, jacobi = false, label_files = false;
	int collective_timeout_s = 600;
	bool sync_io = false;
	bool host_rescale = false;
	int views_in_flight = 0;               // --views-in-flight N: that many views of a pass at once (default 2) where the order allows it and the level is small; 1 = never
	long long in_flight_pixels = 2 << 20;  // ... "small" = at most this many pixels (--in-flight-pixels)
	std::string job, transport = "rccl";
	bool fusion_on_host = false;           // --fusion-on host: RunFusion on the host's cores instead of the GPU (same .ply)
	std::string fusion_kind = "eth";   // eth | tat-intermediate | tat-advanced (APD.h:52-54; the reference's main calls the first)
};

// pair.txt (written by colmap2mvsnet.py:442-448; read at main.cpp:127-170): a whitespace-separated
// stream — the number of views, then per view its id, the number of candidates k and k (id, score)
// pairs.  Candidates with a non-positive score are dropped; at most `max_src` are kept (0 = all).
std::vector<Problem> ReadViewGraph(const path& dense_folder, int max_src) {
	std::vector<Problem> problems;
	std::ifstream in(dense_folder / "pair.txt");
	int n_views = 0;
	if (!(in >> n_views)) return problems;
	for (int i = 0; i < n_views; ++i) {
		Problem p;
		int k = 0;
		if (!(in >> p.ref_image_id >> k)) break;
		p.index = i;
		p.dense_folder = dense_folder;
		p.result_folder = dense_folder / "APD" / ToFormatIndex(p.ref_image_id);
		for (int j = 0; j < k; ++j) {
			int id = 0;
			float score = 0.0f;
			in >> id >> score;
			if (score > 0.0f && (max_src <= 0 || (int)p.src_image_ids.size() < max_src)) p.src_image_ids.push_back(id);
		}
		std::filesystem::create_directories(p.result_folder);
		problems.push_back(std::move(p));
	}
	return problems;
}

int PyramidLevels(const std::vector<Problem>& problems) {   // main.cpp:248-264
	if (problems.empty()) return 0;
	const Mat first = APD::DecodedGray(problems[0].dense_folder / "images" / (ToFormatIndex(problems[0].ref_image_id) + ".jpg"));
	if (first.empty()) return 0;
	int levels = 1;
	for (int side = std::max(first.cols, first.rows); side > 800; side /= 2) ++levels;
	return levels;
}

// one pass of the schedule
struct Pass {
	int level;        // pyramid level (0 = coarsest)
	int scale;        // down-sampling factor of that level
	int geom_index;   // -1: the A pass, else the REFINE_ITER pass number
};

void ConfigurePass(Problem& problem, const Pass& pass, int iteration, int iters, int round_num) {   // main.cpp:457-478, 488-502
	PatchMatchParams& q = problem.params;
	problem.iteration = iteration;
	problem.scale_size = pass.scale;
	q.max_iterations = iters;
	if (pass.level > 0 || pass.geom_index >= 0) {
		q.ransac_threshold = (float)(0.01 - pass.level * 0.00125);
		q.rotate_time = std::min(static_cast<int>(std::pow(2, pass.level)), 4);
	}
	if (pass.geom_index < 0) {
		q.state = pass.level == 0 ? FIRST_INIT : REFINE_INIT;
		q.use_APD = pass.level != 0;
		// main.cpp:465-470: on below the finest pyramid level only (the reference never runs the finest one; with
		// --min-scale 1 its own rule, i < round_num - 1, leaves use_detail off there)
		if (pass.level != 0) q.use_detail = pass.level < round_num - 1;
		q.geom_consistency = false;
		q.weak_peak_radius = 6;
	} else {
		q.state = REFINE_ITER;
		q.use_APD = pass.level != 0;
		q.geom_consistency = true;
		q.weak_peak_radius = std::max(4 - 2 * pass.geom_index, 2);
	}
}

// View -> rank.  Round 4 dealt the views round-robin (v % world).  Views differ in cost — by their pixel count when the images
// of a folder differ in size (APD.cpp:1071-1079 allows it), by their WEAK share (one WEAK pixel costs ~4 others) —, and a pass
// ends when its busiest rank does: longest-predicted-first instead (the view with the largest predicted cost goes to the rank
// with the least load so far; ties: lower view index, lower rank).  Predicted cost = the view's pixels (from the file header:
// every rank computes the same table without decoding anything); the assignment holds for the whole job, so a view's
// maps of the previous pass stay in its owner's result cache.  Equal-size views: the same table as v % world.
// tools/scale_sim.py predicts what the policy is worth.
std::vector<int> AssignViews(const std::vector<Problem>& problems, int world) {
	std::vector<int> owner(problems.size(), 0);
	if (world <= 1) return owner;
	std::vector<std::pair<long long, int>> cost;   // (-pixels, view): ascending sort = largest first, lower index first
	for (const Problem& p : problems) {
		int w = 0, h = 0;
		if (!ImageFileSize(p.dense_folder / "images" / (ToFormatIndex(p.ref_image_id) + ".jpg"), &w, &h)) { w = 1; h = 1; }   // unreadable: the owner will stop the job (DvpFatal) when it loads it
		cost.emplace_back(-(long long)w * h, p.index);
	}
	std::sort(cost.begin(), cost.end());
	std::vector<long long> load((size_t)world, 0);
	for (const auto& c : cost) {
		int best = 0;
		for (int r = 1; r < world; ++r)
			if (load[(size_t)r] < load[(size_t)best]) best = r;
		owner[(size_t)c.second] = best;
		load[(size_t)best] += -c.first;
	}
	return owner;
}

struct ViewResult { Mat depth; };
// The reference writes the ACMM-format maps of a view — depths_geom.dmb + normals.dmb, what ACMM-style fusers read — when
// `problem.iteration == 15` (main.cpp:378-385): the last pass of its default schedule (4 levels x (1 + 3) passes).  Here: at
// that literal index too, and at the last pass of whatever plan the driver was given (--min-scale / --geom-passes shorten it).
int g_final_iteration = 15;
bool g_device_maps = true;   // false (--sync-io / --host-rescale): planes are downloaded and unpacked on the host   // what the exchange step needs from a finished view

// `resident_depth(w, h)`: where on the device the view's new depth map goes for the views that read it as a source (or null)
ViewResult ProcessProblem(const Problem& problem, const std::function<float*(int, int)>& resident_depth = nullptr) {   // main.cpp:267-419
	ViewLog() << "Processing image: " << std::setw(8) << std::setfill('0') << problem.ref_image_id << "..." << std::endl;
	ViewLog() << "Iteration: " << problem.iteration << std::endl;
	const auto start = std::chrono::steady_clock::now();
	// DVP_HOST_TIMING=1: wall time of every host step of the view (where the non-GPU time goes)
	static const bool host_timing = std::getenv("DVP_HOST_TIMING") != nullptr;
	auto lap_t = start;
	auto lap = [&](const char* what) {
		if (!host_timing) return;
		const auto now = std::chrono::steady_clock::now();
		ViewLog() << "  [host] " << what << ": " << std::chrono::duration_cast<std::chrono::microseconds>(now - lap_t).count() / 1000.0 << " ms" << std::endl;
		lap_t = now;
	};
	APD APD(problem);
	APD.InuputInitialization();
	lap("InuputInitialization (images, cameras, previous results)");
	APD.SupportInitialization();
	lap("SupportInitialization (edges, labels, radius)");
	APD.CudaSpaceInitialization();
	lap("CudaSpaceInitialization (context + uploads)");
	APD.SetDataPassHelperInCuda();
	const int width = APD.GetWidth(), height = APD.GetHeight();
	const int nsrc = (int)problem.src_image_ids.size();
	const float dmin = APD.GetDepthMin(), dmax = APD.GetDepthMax();
	Mat depth, normal;
	// Device maps: the unpack loop below is done by the engine, and the maps travel to the host in the background job while
	// the context is already on the next view (25 B per pixel: 27 ms of the 34 ms a full-size view spent outside its kernels)
	std::function<void()> fetch_maps;
	if (g_device_maps) fetch_maps = APD.RunPatchMatchAndStageMaps(depth, normal, resident_depth ? resident_depth(width, height) : nullptr);
	else {
		APD.RunPatchMatch();
		depth = Mat(height, width, CV_32FC1);
		normal = Mat(height, width, CV_32FC3);
	}
	lap(g_device_maps ? "RunPatchMatch" : "RunPatchMatch + download");
	Mat pixel_states = APD.GetPixelStates();
	Mat views = APD.GetSelectedViews();
	Mat radius = problem.params.use_radius ? APD.GetRadiusMap() : Mat();
	// planes -> depth / normal maps; depths outside the admissible range are dropped and the pixel loses
	// its state (main.cpp:300-309)
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (int r = 0; r < (g_device_maps ? 0 : height); ++r) {
		float* z = depth.ptr<float>(r);
		Vec3f* n = normal.ptr<Vec3f>(r);
		uint8_t* st = pixel_states.ptr<uint8_t>(r);
		for (int c = 0; c < width; ++c) {
			const float4 ph = APD.GetPlaneHypothesis(r, c);
			n[c] = Vec3f{ ph.x, ph.y, ph.z };
			const bool usable = !(ph.w < dmin || ph.w > dmax);
			z[c] = usable ? ph.w : 0.0f;
			if (!usable) st[c] = UNKNOWN;
		}
	}
	lap("unpack planes");
	// The depth map is what OTHER views read next (geometric term of their pass): it is published now.  Everything else
	// of this view — the visibility clean-up and the remaining four files — is only read by this view's own next pass:
	// it runs in the background while the GPU is already on the next view (store.cpp).
	const path folder = problem.result_folder;
	if (fetch_maps) ExpectResult(folder / "depths.dmb");   // (a reader of the file waits for the job below; same-size views take the device copy)
	else PublishResult(folder / "depths.dmb", depth);
	for (const char* name : { "APD_normals.dmb", "weak.bin", "selected_views.bin" }) ExpectResult(folder / name);
	if (problem.params.use_radius) ExpectResult(folder / "radius.bin");
	const int scale_size = problem.scale_size;
	const int iteration = problem.iteration;
	const bool timing = host_timing;
	RunInBackground([=]() mutable {
		const auto t0 = std::chrono::steady_clock::now();
		if (fetch_maps) {
			fetch_maps();
			PublishResult(folder / "depths.dmb", depth);
			if (timing) {   // (one write per line: the driver threads print their views' blocks meanwhile)
				std::ostringstream line;
				line << "  [background] maps to the host: " << std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0 << " ms\n";
				std::cout << line.str() << std::flush;
			}
		}
		// Visibility clean-up (main.cpp:311-363): per source view, every 4-connected region of pixels that do
		// NOT select the view and is smaller than 20 * (8 / scale)^2 pixels is switched to "selected".
		const int min_region = 20 * (8 / scale_size) * (8 / scale_size);
		std::vector<Mat> fill(nsrc);   // per source: 1 where the bit has to be set
#pragma omp parallel for schedule(dynamic, 1) num_threads(nsrc < 8 ? (nsrc > 0 ? nsrc : 1) : 8)
		for (int i = 0; i < nsrc; ++i) {
			Mat visible(height, width, CV_8UC1);
			for (int r = 0; r < height; ++r) {
				const uint32_t* w = views.ptr<uint32_t>(r);
				uint8_t* v = visible.ptr<uint8_t>(r);
				for (int c = 0; c < width; ++c) v[c] = ((w[c] >> i) & 1u) ? 255 : 0;
			}
			Mat region(height, width, CV_32S);
			std::vector<int> region_size;
			Connect(visible, region, region_size);
			Label_Update(region, region_size);
			fill[i] = Mat(height, width, CV_8UC1);
			for (int r = 0; r < height; ++r) {
				const int* lab = region.ptr<int>(r);
				uint8_t* f = fill[i].ptr<uint8_t>(r);
				for (int c = 0; c < width; ++c) f[c] = (lab[c] != 0 && region_size[lab[c]] >= min_region) ? 0 : 1;
			}
		}
#pragma omp parallel for schedule(static) num_threads(HostThreads())
		for (int r = 0; r < height; ++r) {
			uint32_t* w = views.ptr<uint32_t>(r);
			for (int c = 0; c < width; ++c) {
				unsigned int mask = 0;
				for (int i = 0; i < nsrc; ++i) mask |= (unsigned int)fill[i].at<uint8_t>(r, c) << i;
				w[c] = mask;                      // APD.SetPixelSelectedViews(r, c, mask)
			}
		}
		PublishResult(folder / "APD_normals.dmb", normal);
		PublishResult(folder / "weak.bin", pixel_states);
		PublishResult(folder / "selected_views.bin", views);
		if (!radius.empty()) PublishResult(folder / "radius.bin", radius);
		if (iteration == 15 || iteration == g_final_iteration) {   // main.cpp:378-382 (weak.png is a debug image: not written)
			writeDepthDmb(folder / "depths_geom.dmb", depth);
			writeNormalDmb(folder / "normals.dmb", normal);
		}
		if (timing) {
			std::ostringstream line;
			line << "  [background] visibility-mask clean-up + publish: " << std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0 << " ms\n";
			std::cout << line.str() << std::flush;
		}
	});
	lap("hand-over to the background worker");
	const auto end = std::chrono::steady_clock::now();
	const DvpTimings& t = APD.GetTimings();
	ViewLog() << "Processing image: " << std::setw(8) << std::setfill('0') << problem.ref_image_id << " done!" << std::endl;
	if (host_timing) {   // where the GPU time of the view went, launch site by launch site (event pairs around each site: gaps between sites — a host that is late with the next launch — show as the difference to the total)
		static const char* names[DVP_ST_COUNT] = { "gen_edge_inform", "find_nearest_strong", "gen_neighbours", "neighbour_update", "random_init", "strong_update", "ransac_fit", "weak_update", "get_depth_normal", "filter_strong", "depth_to_weak", "local_refine", "strong_prep" };
		double sum = 0.0;
		ViewLog() << "  [gpu]";
		for (int i = 0; i < DVP_ST_COUNT; ++i)
			if (t.stage_ms[i] > 0.0) { ViewLog() << " " << names[i] << " " << t.stage_ms[i]; sum += t.stage_ms[i]; }
		ViewLog() << " | sites " << sum << " of total " << t.total_ms << " ms" << std::endl;
	}
	ViewLog() << "Cost time: " << std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count() << " ms (GPU RunPatchMatch "
	          << t.total_ms << " ms, " << (double)width * height * problem.params.max_iterations / (t.total_ms * 1e3) << " Mpx/s/iter)" << std::endl;
	return ViewResult{ depth };
}

// depth maps of every view, resident on this rank's device, refreshed after each pass.  Views may differ in size
// (the reference pads / crops source images and rescales source depth maps, APD.cpp:1071-1079, 1147-1166): every
// map travels with its own dimensions.
class DepthExchange {
public:
	DepthExchange(RankComm& comm, const std::vector<Problem>& problems, const std::vector<int>& owner) : comm_(comm), problems_(problems), owner_(owner) {}
	~DepthExchange() { Release(); }
	// `mine`: view index -> the depth map this rank computed in the pass just finished
	void Publish(const std::map<int, Mat>& mine) {
		// agreed check first: a rank that lacks one of its maps must not leave the others inside a collective
		bool ok = true;
		for (size_t v = 0; v < problems_.size(); ++v)
			if (owner_[v] == comm_.rank() && (mine.find((int)v) == mine.end() || mine.at((int)v).empty())) ok = false;
		if (!comm_.AllOk(ok)) {
			std::cerr << "DepthExchange: a rank is missing a depth map of the pass" << std::endl;
			if (comm_.world() > 1) comm_.Abort("DepthExchange: missing depth map");
			exit(EXIT_FAILURE);
		}
		for (size_t v = 0; v < problems_.size(); ++v) {
			const int owner = owner_[v];
			int dims[2] = { 0, 0 };
			if (owner == comm_.rank()) { dims[0] = mine.at((int)v).cols; dims[1] = mine.at((int)v).rows; }
			comm_.BroadcastHost(dims, sizeof(dims), owner);
			Slot& s = maps_[problems_[v].ref_image_id];
			const size_t count = (size_t)dims[0] * dims[1];
			if (s.w != dims[0] || s.h != dims[1]) {
				RankComm::DeviceFree(s.dev);
				s.dev = RankComm::DeviceAlloc(count);
				s.w = dims[0];
				s.h = dims[1];
			}
			if (owner == comm_.rank()) RankComm::HostToDevice(s.dev, mine.at((int)v).ptr<float>(0), count);
			comm_.BroadcastDevice(s.dev, count, owner);
			APD::SetResidentDepth(problems_[v].ref_image_id, s.dev, s.w, s.h);
		}
	}
	void Release() {
		APD::ClearResidentDepths();
		for (auto& kv : maps_) RankComm::DeviceFree(kv.second.dev);
		maps_.clear();
	}
private:
	struct Slot { float* dev = nullptr; int w = 0, h = 0; };
	RankComm& comm_;
	const std::vector<Problem>& problems_;
	const std::vector<int>& owner_;
	std::map<int, Slot> maps_;
};

// Single-rank, in-place order (the reference's): a finished view's depth map goes to the device right away and stays
// there, so that the views after it — which see it as a source map in the same pass, exactly as they would through
// APD/<id>/depths.dmb — take it with a device-to-device copy instead of a 100 MB file read + upload per source.
class InPlaceDepths {
public:
	~InPlaceDepths() { Release(); }
	void Update(int image_id, const Mat& depth) {
		Slot& s = maps_[image_id];
		if (s.w != depth.cols || s.h != depth.rows) {
			APD::UnsetResidentDepth(image_id);   // the registry must not name a freed block, not even until the new one is registered
			RankComm::DeviceFree(s.dev);
			s.dev = RankComm::DeviceAlloc((size_t)depth.cols * depth.rows);
			s.w = depth.cols;
			s.h = depth.rows;
		}
		RankComm::HostToDevice(s.dev, depth.ptr<float>(0), (size_t)s.w * s.h);
		APD::SetResidentDepth(image_id, s.dev, s.w, s.h);
	}
	// ... or straight from the engine's staged maps (dvp_download_maps_begin): the slot the view's new map is copied to on
	// the device, registered with Commit once it is there
	float* Reserve(int image_id, int w, int h) {
		Slot& s = maps_[image_id];
		if (s.w != w || s.h != h) {
			APD::UnsetResidentDepth(image_id);   // ... until Commit
			RankComm::DeviceFree(s.dev);
			s.dev = RankComm::DeviceAlloc((size_t)w * h);
			s.w = w;
			s.h = h;
		}
		return s.dev;
	}
	void Commit(int image_id) {
		const Slot& s = maps_[image_id];
		if (s.dev) APD::SetResidentDepth(image_id, s.dev, s.w, s.h);
	}
	void Release() {
		APD::ClearResidentDepths();
		for (auto& kv : maps_) RankComm::DeviceFree(kv.second.dev);
		maps_.clear();
	}
private:
	struct Slot { float* dev = nullptr; int w = 0, h = 0; };
	std::map<int, Slot> maps_;
};

// The float images of a pyramid level resident on the device: every image this rank's views use (as reference or as
// source) is uploaded ONCE per level; a view then takes its ten planes with device-to-device copies instead of the host
// re-sending 100 MB per image and per view that uses it (20 of the 50 ms of host time per full-resolution view).  Bounded
// by DVP_RESIDENT_IMAGES_GB (default 24): what does not fit keeps the host path.  Images whose size differs from the
// view's (padded / cropped sources) are simply not matched by APD::CudaSpaceInitialization.
class LevelImages {
public:
	~LevelImages() { Release(); }
	void Fill(const std::vector<Problem*>& owned, int scale) {
		Release();
		const char* e = std::getenv("DVP_RESIDENT_IMAGES_GB");
		const size_t budget = (size_t)(e ? std::max(0, std::atoi(e)) : 24) << 30;
		size_t used = 0;
		std::set<int> ids;
		for (const Problem* p : owned) { ids.insert(p->ref_image_id); ids.insert(p->src_image_ids.begin(), p->src_image_ids.end()); }
		if (owned.empty()) return;
		Problem q = *owned[0];
		q.scale_size = scale;
		for (int id : ids) {
			int oc = 0, orr = 0;
			const Mat img = APD::CachedImage(q, id, &oc, &orr);   // reference role: the image at its own size
			const size_t count = (size_t)img.cols * img.rows;
			if (img.empty() || used + count * 4 > budget) continue;
			float* dev = RankComm::DeviceAlloc(count);
			RankComm::HostToDevice(dev, img.ptr<float>(0), count);
			blocks_.push_back(dev);
			used += count * 4;
			APD::SetResidentImage(id, scale, dev, img.cols, img.rows, oc, orr);
		}
	}
	void Release() {
		APD::ClearResidentImages();
		for (float* b : blocks_) RankComm::DeviceFree(b);
		blocks_.clear();
	}
private:
	std::vector<float*> blocks_;
};

// The owner of a view (rank v % world) decodes + resizes its image at this level and broadcasts it; the others
// take it from the broadcast into their image cache: every image file is read by exactly one process and the
// decode work is spread over the ranks.
void ShareLevelImages(RankComm& comm, std::vector<Problem>& problems, const std::vector<int>& owner_of, int scale) {
	if (comm.world() <= 1) return;
	APD::ReserveImageCache(problems.size());
	for (Problem& p : problems) {
		p.scale_size = scale;
		const int owner = owner_of[(size_t)p.index];
		int meta[4] = { 0, 0, 0, 0 };   // cols, rows, original cols, original rows
		Mat img;
		if (comm.rank() == owner) {
			img = APD::CachedImage(p, p.ref_image_id, &meta[2], &meta[3]);
			meta[0] = img.cols;
			meta[1] = img.rows;
		}
		comm.BroadcastHost(meta, sizeof(meta), owner);
		if (comm.rank() != owner) img = Mat(meta[1], meta[0], CV_32FC1);
		comm.BroadcastHost(img.data, (size_t)meta[0] * meta[1] * sizeof(float), owner);
		if (comm.rank() != owner) APD::InsertCachedImage(p, p.ref_image_id, img, meta[2], meta[3]);
	}
}

Options ParseOptions(int argc, char** argv) {
	Options o;
	o.dense_folder = argv[1];
	int a = 2;
	if (argc > 2 && argv[2][0] != '-') { o.gpu = std::atoi(argv[2]); a = 3; }
	for (; a < argc; ++a) {
		const std::string s = argv[a];
		auto val = [&]() { return (a + 1 < argc) ? std::atoll(argv[++a]) : 0ll; };
		if (s == "--max-src") o.max_src = (int)val();
		else if (s == "--iters") o.iters = (int)val();              // the reference hard-wires 3 (main.cpp:476,502)
		else if (s == "--min-scale") o.min_scale = (int)val();      // the reference stops at half resolution (main.cpp:450,456)
		else if (s == "--passes") o.geom_passes = (int)val();       // REFINE_ITER passes per level (3 in the reference)
		else if (s == "--seed") o.seed = (uint64_t)val();
		else if (s == "--rank") o.rank = (int)val();
		else if (s == "--world") o.world = (int)val();
		else if (s == "--job") { if (a + 1 < argc) o.job = argv[++a]; }
		else if (s == "--transport") { if (a + 1 < argc) o.transport = argv[++a]; }
		else if (s == "--collective-timeout") o.collective_timeout_s = (int)val();
		else if (s == "--jacobi") o.jacobi = true;
		else if (s == "--labels") o.label_files = true;          // load labels_<s>.dmb (APD::SetUseLabelFiles)
		else if (s == "--no-fusion") o.fusion = false;
		else if (s == "--sync-io") o.sync_io = true;               // no result cache / background worker / device rescale: the reference's synchronous file flow
		else if (s == "--views-in-flight") { if (a + 1 < argc) o.views_in_flight = std::min(APD::kMaxViewsInFlight, std::max(1, atoi(argv[++a]))); }   // one pooled engine context per view in flight
		else if (s == "--in-flight-pixels") { if (a + 1 < argc) o.in_flight_pixels = atoll(argv[++a]); }
		else if (s == "--host-rescale") o.host_rescale = true;     // the coarser level's maps are up-sampled on the host (APD::SetDeviceRescale(false))
		else if (s == "--fusion") { if (a + 1 < argc) o.fusion_kind = argv[++a]; }
		else if (s == "--fusion-on") { if (a + 1 < argc) o.fusion_on_host = std::string(argv[++a]) == "host"; }   // device (default) | host
	}
	if (o.world > 1) o.jacobi = true;
	if (o.job.empty())   // a launcher-provided id; with --world > 1 RankComm refuses to start without one
		for (const char* e : { "DVP_JOB_ID", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID" })
			if (const char* v = std::getenv(e)) { o.job = v; break; }
	return o;
}

}  // namespace

int main(int argc, char** argv) {
	if (argc < 2) {
		std::cerr << "USAGE: apd dense_folder [gpu_index] [--max-src N] [--iters N] [--min-scale S] [--passes P] [--seed X] [--rank R --world N [--job ID]] [--jacobi] [--no-fusion | --fusion KIND] [--fusion-on device|host] [--views-in-flight N]\n";
		return EXIT_FAILURE;
	}
	const Options opt = ParseOptions(argc, argv);
	const bool main_timing = std::getenv("DVP_HOST_TIMING") != nullptr;
	auto main_t = std::chrono::steady_clock::now();
	auto main_lap = [&](const std::string& what) {   // DVP_HOST_TIMING: what the driver does between the passes
		if (!main_timing) return;
		const auto now = std::chrono::steady_clock::now();
		std::cout << "[main] " << what << ": " << std::chrono::duration_cast<std::chrono::microseconds>(now - main_t).count() / 1000.0 << " ms" << std::endl;
		main_t = now;
	};
	std::filesystem::create_directories(opt.dense_folder / "APD");
	// The engine's sweep passes through a band of rows of at most 24 GB of cost records (the whole 25-Mpx view with 9 sources: 67 GB =
	// 2.3 s of fresh device memory, part of which the views running meanwhile wait for; 3 bands cost a geometric full-size pass 10 ms
	// per view in drained launches, 5 bands 20): a full-size context is 64 GB instead of 107.  Measured neutral on the ten-view
	// schedule (32.8 s either way).  DVP_SWEEP_BAND_GB in the environment wins; 0 = whole image.
	setenv("DVP_SWEEP_BAND_GB", "24", 0);
	SetHostThreadShare(opt.world);
	if (opt.world > 1) std::cout << "rank " << opt.rank << " of " << opt.world << ": " << HostThreads() << " host threads (of " << std::thread::hardware_concurrency() << " cores)" << std::endl;
	APD::SetDevice(opt.gpu);
	{   // before any thread is started and any large block is touched
		const int node = RankComm::BindProcessNearDevice(opt.gpu);
		if (node >= 0) std::cout << "GPU " << opt.gpu << " hangs off NUMA node " << node << ": process confined to its CPUs" << std::endl;
	}
	APD::SetSeed(opt.seed);
	APD::SetUseLabelFiles(opt.label_files);
	SetResultCache(!opt.sync_io);
	APD::SetDeviceRescale(!opt.sync_io && !opt.host_rescale);
	g_device_maps = !opt.sync_io && !opt.host_rescale;
	RankComm comm(opt.rank, opt.world, opt.gpu, (opt.dense_folder / "APD" / ".rccl_id").string(), opt.job, opt.collective_timeout_s, opt.transport);
	// every print-and-exit of the host library (unreadable image, missing weak.bin, engine error ...) becomes an agreed
	// abort of the whole job when there are peers
	DvpSetFatalHook([](const char* msg) { if (RankComm* c = RankComm::Current()) if (c->world() > 1) c->Abort(msg); });

	std::vector<Problem> problems = ReadViewGraph(opt.dense_folder, opt.max_src);
	std::cout << "There are " << problems.size() << " problems needed to be processed!" << std::endl;
	const std::vector<int> owner_of = AssignViews(problems, opt.world);
	if (opt.world > 1) {
		std::cout << "rank " << opt.rank << " owns views";
		for (const Problem& p : problems) if (owner_of[(size_t)p.index] == opt.rank) std::cout << " " << p.index;
		std::cout << std::endl;
	}
	{   // this rank's images are decoded in the background from now on (every pyramid level is made from the decoded file)
		std::vector<path> mine;
		for (const Problem& p : problems)
			if (owner_of[(size_t)p.index] == opt.rank) mine.push_back(opt.dense_folder / "images" / (ToFormatIndex(p.ref_image_id) + ".jpg"));
		if (opt.world == 1)   // a single rank also reads every source image itself
			for (const Problem& p : problems)
				for (int id : p.src_image_ids) mine.push_back(opt.dense_folder / "images" / (ToFormatIndex(id) + ".jpg"));
		APD::PrefetchDecoded(mine);
	}
	main_lap("start-up (options, rendezvous, view graph, decode prefetch started)");
	const int round_num = PyramidLevels(problems);
	main_lap("PyramidLevels (first image decoded)");
	std::cout << "Round nums: " << round_num << std::endl;

	// the pass list: level i has scale 2^(round_num-1-i); levels run while the scale is >= min_scale
	// (min_scale = 2 reproduces `i < round_num - 1`, main.cpp:450; a single-level pyramid always runs)
	std::vector<Pass> plan;
	for (int level = 0; level < round_num; ++level) {
		const int scale = 1 << (round_num - 1 - level);
		if (scale < opt.min_scale && round_num != 1) break;
		plan.push_back(Pass{ level, scale, -1 });
		for (int j = 0; j < opt.geom_passes; ++j) plan.push_back(Pass{ level, scale, j });
	}

	g_final_iteration = (int)plan.size() - 1;

	std::unique_ptr<DepthExchange> exchange;
	std::unique_ptr<InPlaceDepths> inplace;
	if (opt.jacobi) { exchange.reset(new DepthExchange(comm, problems, owner_of)); APD::SetResidentDownloader(&RankComm::DeviceToHost); }
	else if (!opt.sync_io) inplace.reset(new InPlaceDepths());
	int shared_scale = -1;
	LevelImages level_images;
	for (size_t it = 0; it < plan.size(); ++it) {
		const Pass& pass = plan[it];
		const bool new_level = pass.scale != shared_scale;
		main_t = std::chrono::steady_clock::now();
		if (new_level) {
			ShareLevelImages(comm, problems, owner_of, pass.scale);
			shared_scale = pass.scale;
			if (exchange) exchange->Release();   // maps of the coarser level do not fit this one (and its A pass has no geometric term)
		}
		std::map<int, Mat> mine;
		std::vector<Problem*> owned;
		for (Problem& problem : problems) {
			ConfigurePass(problem, pass, (int)it, opt.iters, round_num);
			if (owner_of[(size_t)problem.index] == opt.rank) owned.push_back(&problem);
		}
		if (new_level && !opt.sync_io) level_images.Fill(owned, pass.scale);
		if (new_level) main_lap("level " + std::to_string(pass.level) + ": images shared + resident on the device");
		// last pass of a level: the next level's context and float images are made by helper threads while the GPU works
		if (!opt.sync_io && it + 1 < plan.size() && plan[it + 1].scale != pass.scale && !owned.empty()) {
			int nw = 0, nh = 0;
			if (APD::LevelSize(*owned[0], plan[it + 1].scale, &nw, &nh)) APD::PrewarmContext(nw, nh, (int)owned[0]->src_image_ids.size() + 1);
			std::vector<Problem> mine_next;
			for (const Problem* p : owned) mine_next.push_back(*p);
			APD::PrefetchLevelImages(mine_next, plan[it + 1].scale);
		}
		// The edge / label maps of an A pass (main.cpp:480: GetProblemEdges before every view).  Round 4 made the NEXT view's maps
		// while the GPU worked on the current one — on the one background worker, behind that view's clean-up job: at the coarse
		// levels a view's kernels take 20-70 ms, its label map (Roberts cross + components + Hough lines on the FULL-size image)
		// longer, and SupportInitialization waited 32 of 59 ms per view (profiles/r04g_e2e_apd.txt, pass 0).  Now every owned
		// view's maps are announced at the start of the pass and made by a few helper threads in view order; a view only waits
		// if its own maps are not there yet (LoadResult waits for announced files).  Same files (store.cpp).
		std::vector<std::future<void>> edge_jobs;
		auto start_edge_jobs = [&edge_jobs](const std::vector<Problem>& views) {
			struct Queue { std::mutex m; size_t next = 0; std::vector<std::pair<Problem, std::vector<path>>> items; };
			auto q = std::make_shared<Queue>();
			for (const Problem& p : views) {
				std::vector<path> outs = ProblemEdgeOutputs(p);   // (what exists or is announced already is not made again)
				for (const path& f : outs) ExpectResult(f);
				if (!outs.empty()) q->items.emplace_back(p, std::move(outs));
			}
			const int helpers = (int)std::min<size_t>(q->items.size(), (size_t)std::max(1, std::min(4, HostThreads() / 4)));
			for (int h = 0; h < helpers; ++h)
				edge_jobs.push_back(std::async(std::launch::async, [q]() {
					// two threads per helper: with full-size teams the helpers starved the thread that issues the launches — the GPU
					// idled between kernels (RunPatchMatch of a 1552x1032 view: 67 -> 131 ms)
					SetThisThreadHostThreads(2);
					for (;;) {
						size_t i;
						{ std::lock_guard<std::mutex> lk(q->m); i = q->next++; }
						if (i >= q->items.size()) return;
						GetProblemEdges(q->items[i].first, q->items[i].second);
					}
				}));
		};
		if (!opt.sync_io) {
			std::vector<Problem> views;
			if (pass.geom_index < 0)
				for (const Problem* p : owned) views.push_back(*p);
			// ... and during the last pass of a level, the maps of the NEXT level's A pass (its first view used to wait for its
			// own label map: 0.6 s at full size)
			if (it + 1 < plan.size() && plan[it + 1].scale != pass.scale && plan[it + 1].geom_index < 0)
				for (const Problem* p : owned) {
					Problem q = *p;
					ConfigurePass(q, plan[it + 1], (int)it + 1, opt.iters, round_num);
					views.push_back(q);
				}
			if (!views.empty()) start_edge_jobs(views);
		}
		// Views in flight.  The reference visits the views of a pass one after the other (main.cpp:479-486) and a view of a
		// geometric pass reads the maps the views before it have just written — that order is kept.  Where no view reads another
		// view's result of the SAME pass — a photometric pass, or any pass with the depth exchange (--jacobi / several ranks: maps of
		// the previous pass) — and the level is small, two or more views run at once, each on its own engine context and stream:
		// a 776x516 view is 6 k waves for a 256-CU part and 20 ms of kernels between 10 ms of host work (profiles/r04g_e2e_apd.txt).
		// Same files either way (every view's inputs are fixed before the pass).
		int in_flight = 1;
		if (!opt.sync_io && (pass.geom_index < 0 || exchange) && !owned.empty()) {
			int lw = 0, lh = 0;
			if (APD::LevelSize(*owned[0], pass.scale, &lw, &lh) && (size_t)lw * lh <= (size_t)opt.in_flight_pixels) in_flight = opt.views_in_flight > 0 ? opt.views_in_flight : 2;
			in_flight = (int)std::min<size_t>((size_t)std::max(1, in_flight), owned.size());
		}
		// the fusion's colour images are decoded while the last pass runs (rank 0 fuses)
		if (it + 1 == plan.size() && opt.fusion && opt.rank == 0 && !opt.sync_io) PrefetchFusionImages(opt.dense_folder, problems);
		main_lap("pass " + std::to_string(it) + ": helpers started");
		const auto pass_t0 = std::chrono::steady_clock::now();
		if (in_flight <= 1) {
			for (size_t k = 0; k < owned.size(); ++k) {
				Problem& problem = *owned[k];
				if (pass.geom_index < 0 && opt.sync_io) GetProblemEdges(problem);   // main.cpp:480, synchronously
				bool on_device = false;
				ViewResult r = ProcessProblem(problem, [&](int w, int h) -> float* {
					if (!inplace || !g_device_maps) return nullptr;
					on_device = true;
					return inplace->Reserve(problem.ref_image_id, w, h);
				});
				if (exchange) mine[problem.index] = r.depth;
				if (inplace && on_device) inplace->Commit(problem.ref_image_id);
				else if (inplace) inplace->Update(problem.ref_image_id, r.depth);
			}
		} else {
			std::atomic<size_t> next{0};
			std::mutex results;   // `mine`, the in-place maps, the order of the views' log blocks
			std::vector<std::thread> workers;
			const int team = std::max(2, HostThreads() / in_flight);
			for (int wkr = 0; wkr < in_flight; ++wkr)
				workers.emplace_back([&, team]() {
					RankComm::BindThisThread(opt.gpu);   // HIP's current device is per thread
					SetThisThreadHostThreads(team);
					for (;;) {
						const size_t k = next++;
						if (k >= owned.size()) return;
						Problem& problem = *owned[k];
						std::ostringstream log;
						SetViewLog(&log);
						bool on_device = false;
						ViewResult r = ProcessProblem(problem, [&](int w, int h) -> float* {
							if (!inplace || !g_device_maps) return nullptr;
							std::lock_guard<std::mutex> lk(results);
							on_device = true;
							return inplace->Reserve(problem.ref_image_id, w, h);
						});
						SetViewLog(nullptr);
						std::lock_guard<std::mutex> lk(results);
						std::cout << log.str() << std::flush;
						if (exchange) mine[problem.index] = r.depth;
						if (inplace && on_device) inplace->Commit(problem.ref_image_id);
						else if (inplace) inplace->Update(problem.ref_image_id, r.depth);
					}
				});
			for (std::thread& t : workers) t.join();
		}
		if (std::getenv("DVP_HOST_TIMING")) {   // (one write: the background jobs of the pass' views may still be printing their own timing lines)
			std::ostringstream line;
			line << "Pass " << it << ": " << owned.size() << " views in " << std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - pass_t0).count() / 1000.0
			     << " ms, " << in_flight << " in flight\n";
			std::cout << line.str() << std::flush;
		}
		// With peers, a view whose size differs from a source's takes that source's depth map from APD/<id>/depths.dmb
		// (APD.cpp fallback from the resident maps to LoadResult): the owner's background writer must have put this
		// pass' files on disk BEFORE the barrier lets anyone into the next pass, or the reader sees the previous pass'
		// map (or none) depending on timing.  One rank has no other reader: its cache serves its own next pass.
		main_t = std::chrono::steady_clock::now();
		for (auto& j : edge_jobs) j.get();
		if (opt.world > 1) FlushResults();
		if (exchange) WaitBackgroundJobs();      // (the maps of `mine` are filled by the views' background jobs)
		if (exchange) exchange->Publish(mine);   // collective: also the barrier between passes
		else comm.Barrier();
		main_lap("pass " + std::to_string(it) + ": helpers joined, exchange / barrier");
		if (pass.geom_index == opt.geom_passes - 1 || (opt.geom_passes == 0 && pass.geom_index < 0)) std::cout << "Round: " << pass.level << " done\n";
	}
	main_t = std::chrono::steady_clock::now();
	if (exchange) exchange->Release();
	exchange.reset();
	inplace.reset();
	level_images.Release();
	APD::ReleasePooledContext();
	main_lap("contexts and resident maps released");
	// Several ranks: every result file of this rank is on disk before rank 0's fusion reads the folder.  One rank: the fusion takes
	// the maps from the result cache (LoadResult waits for a map whose background job has not published it yet), so the last views'
	// files are written while it runs; ShutdownResultStore waits for them.
	if (opt.world > 1 || opt.sync_io) FlushResults();
	main_lap("FlushResults (background jobs + file writes)");
	comm.Barrier();
	if (opt.fusion && opt.rank == 0) {
		SetFusionOnHost(opt.fusion_on_host);   // all three variants run on the GPU (dvp_fuse_*) unless --fusion-on host
		SetFusionDevice(opt.gpu);
		if (opt.fusion_kind == "tat-intermediate") RunFusion_TAT_Intermediate(opt.dense_folder, problems);
		else if (opt.fusion_kind == "tat-advanced") RunFusion_TAT_advanced(opt.dense_folder, problems);
		else RunFusion(opt.dense_folder, problems);
	}
	ShutdownResultStore();
	main_lap("fusion + shutdown");
	std::cout << "All done\n";
	return EXIT_SUCCESS;
}
