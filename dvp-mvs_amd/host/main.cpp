// main.cpp — driver reproducing the reference's schedule (/root/reference/main.cpp:127-170,
// 248-528) on top of `class APD`, with the knobs BASELINE.json's configs need exposed:
//   apd <dense_folder> [gpu_index] [--max-src N] [--iters N] [--min-scale S] [--passes P]
//       [--seed X] [--rank R --world N] [--no-fusion]
// Views are independent within a pass, so `--rank/--world` shard them round-robin across
// processes (one per GPU); passes are separated by a file-system barrier (the per-view result
// files are the inter-pass API of the reference).
#include "APD.h"
#include <cstdlib>
#include <thread>

static void GenerateSampleList(const path& dense_folder, std::vector<Problem>& problems, int max_src) {   // main.cpp:127-170
	path cluster_list_path = dense_folder / path("pair.txt");
	problems.clear();
	std::ifstream file(cluster_list_path);
	std::stringstream iss;
	std::string line;
	int num_images = 0;
	std::getline(file, line);
	iss.str(line);
	iss >> num_images;
	for (int i = 0; i < num_images; ++i) {
		Problem problem;
		problem.index = i;
		iss.clear();
		std::getline(file, line);
		iss.str(line);
		iss >> problem.ref_image_id;
		problem.dense_folder = dense_folder;
		problem.result_folder = dense_folder / path("APD") / path(ToFormatIndex(problem.ref_image_id));
		std::filesystem::create_directories(problem.result_folder);
		int num_src_images = 0;
		iss.clear();
		std::getline(file, line);
		iss.str(line);
		iss >> num_src_images;
		for (int j = 0; j < num_src_images; ++j) {
			int id;
			float score;
			iss >> id >> score;
			if (score <= 0.0f) continue;
			if (max_src > 0 && (int)problem.src_image_ids.size() >= max_src) continue;   // BASELINE configs use 3/5/9 views
			problem.src_image_ids.push_back(id);
		}
		problems.push_back(problem);
	}
}

static int ComputeRoundNum(const std::vector<Problem>& problems) {   // main.cpp:248-264
	if (problems.empty()) return 0;
	Mat image = ReadImageGray(problems[0].dense_folder / path("images") / path(ToFormatIndex(problems[0].ref_image_id) + ".jpg"));
	if (image.empty()) return 0;
	int max_size = std::max(image.cols, image.rows);
	int round_num = 1;
	while (max_size > 800) { max_size /= 2; round_num++; }
	return round_num;
}

static void setBit_YZL(unsigned int* input, const unsigned int n) { (*input) |= (unsigned int)(1 << n); }

static void ProcessProblem(const Problem& problem) {   // main.cpp:267-419
	std::cout << "Processing image: " << std::setw(8) << std::setfill('0') << problem.ref_image_id << "..." << std::endl;
	std::cout << "Iteration: " << problem.iteration << std::endl;
	auto start = std::chrono::steady_clock::now();
	// DVP_HOST_TIMING=1: wall time of every host step of the view (where the non-GPU time goes)
	static const bool host_timing = std::getenv("DVP_HOST_TIMING") != nullptr;
	auto lap_t = start;
	auto lap = [&](const char* what) {
		if (!host_timing) return;
		const auto now = std::chrono::steady_clock::now();
		std::cout << "  [host] " << what << ": " << std::chrono::duration_cast<std::chrono::microseconds>(now - lap_t).count() / 1000.0 << " ms" << std::endl;
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
	APD.RunPatchMatch();
	lap("RunPatchMatch + download");
	int width = APD.GetWidth(), height = APD.GetHeight();
	Mat depth(height, width, CV_32FC1), normal(height, width, CV_32FC3);
	Mat pixel_states = APD.GetPixelStates();
	const int nsrc = (int)problem.src_image_ids.size();
	std::vector<Mat> vis(nsrc);
	for (int i = 0; i < nsrc; ++i) vis[i] = Mat(height, width, CV_8UC1);
#pragma omp parallel for schedule(static) num_threads(8)
	for (int r = 0; r < height; ++r)
		for (int c = 0; c < width; ++c) {
			float4 ph = APD.GetPlaneHypothesis(r, c);
			depth.at<float>(r, c) = ph.w;
			if (depth.at<float>(r, c) < APD.GetDepthMin() || depth.at<float>(r, c) > APD.GetDepthMax()) {
				depth.at<float>(r, c) = 0;
				pixel_states.at<uint8_t>(r, c) = UNKNOWN;
			}
			normal.at<Vec3f>(r, c) = Vec3f{ph.x, ph.y, ph.z};
			unsigned int views = (unsigned int)APD.GetPixelSelectedViews(r, c);
			for (int i = 0; i < nsrc; ++i) vis[i].at<uint8_t>(r, c) = ((views >> i) & 1) ? 255 : 0;
		}
	lap("unpack planes / view masks");
	// visibility-mask clean-up: invisible components smaller than 20*(8/scale)^2 px become visible
	// (main.cpp:323-363)
	const int thr = 20 * (8 / problem.scale_size) * (8 / problem.scale_size);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nsrc < 8 ? (nsrc > 0 ? nsrc : 1) : 8)   // one source view's mask per thread
	for (int i = 0; i < nsrc; ++i) {
		Mat lab_mask(height, width, CV_32S);
		std::vector<int> label_cnt;
		Connect(vis[i], lab_mask, label_cnt);
		Label_Update(lab_mask, label_cnt);
		for (int y = 0; y < height; y++)
			for (int x = 0; x < width; x++) {
				const int label = lab_mask.at<int>(y, x);
				const bool big = label != 0 && label_cnt[label] >= thr;
				vis[i].at<uint8_t>(y, x) = big ? 0 : 255;
			}
	}
#pragma omp parallel for schedule(static) num_threads(8)
	for (int y = 0; y < height; y++)
		for (int x = 0; x < width; x++) {
			unsigned int v = 0;
			for (int i = 0; i < nsrc; ++i)
				if (vis[i].at<uint8_t>(y, x) == 255) setBit_YZL(&v, i);
			APD.SetPixelSelectedViews(y, x, (int)v);
		}
	lap("visibility-mask clean-up");
	WriteBinMat(problem.result_folder / path("depths.dmb"), depth);
	WriteBinMat(problem.result_folder / path("APD_normals.dmb"), normal);
	WriteBinMat(problem.result_folder / path("weak.bin"), pixel_states);
	WriteBinMat(problem.result_folder / path("selected_views.bin"), APD.GetSelectedViews());
	if (problem.params.use_radius) WriteBinMat(problem.result_folder / path("radius.bin"), APD.GetRadiusMap());
	lap("write results");
	auto end = std::chrono::steady_clock::now();
	const DvpTimings& t = APD.GetTimings();
	std::cout << "Processing image: " << std::setw(8) << std::setfill('0') << problem.ref_image_id << " done!" << std::endl;
	std::cout << "Cost time: " << std::chrono::duration_cast<std::chrono::milliseconds>(end - start).count() << " ms (GPU RunPatchMatch "
	          << t.total_ms << " ms, " << (double)width * height * problem.params.max_iterations / (t.total_ms * 1e3) << " Mpx/s/iter)" << std::endl;
}

// file-system barrier between passes for multi-process runs
static void PassBarrier(const path& dir, int pass, int rank, int world) {
	if (world <= 1) return;
	std::filesystem::create_directories(dir);
	{ std::ofstream f(dir / ("pass" + std::to_string(pass) + "_rank" + std::to_string(rank))); f << "done\n"; }
	for (int r = 0; r < world; ++r)
		while (!std::filesystem::exists(dir / ("pass" + std::to_string(pass) + "_rank" + std::to_string(r))))
			std::this_thread::sleep_for(std::chrono::milliseconds(20));
}

int main(int argc, char** argv) {
	if (argc < 2) {
		std::cerr << "USAGE: apd dense_folder [gpu_index] [--max-src N] [--iters N] [--min-scale S] [--passes P] [--seed X] [--rank R --world N] [--no-fusion]\n";
		return EXIT_FAILURE;
	}
	path dense_folder(argv[1]);
	int gpu_index = 0, max_src = 0, iters = 3, min_scale = 2, geom_passes = 3, rank = 0, world = 1;
	uint64_t seed = 1234;
	bool fusion = true;
	int a = 2;
	if (argc > 2 && argv[2][0] != '-') { gpu_index = std::atoi(argv[2]); a = 3; }
	for (; a < argc; ++a) {
		std::string s = argv[a];
		auto val = [&]() { return (a + 1 < argc) ? std::atoll(argv[++a]) : 0ll; };
		if (s == "--max-src") max_src = (int)val();
		else if (s == "--iters") iters = (int)val();         // the reference hard-wires 3 (main.cpp:476,502)
		else if (s == "--min-scale") min_scale = (int)val(); // the reference stops at half resolution (main.cpp:450,456)
		else if (s == "--passes") geom_passes = (int)val();  // REFINE_ITER passes per round (3 in the reference)
		else if (s == "--seed") seed = (uint64_t)val();
		else if (s == "--rank") rank = (int)val();
		else if (s == "--world") world = (int)val();
		else if (s == "--no-fusion") fusion = false;
	}
	std::filesystem::create_directories(dense_folder / path("APD"));
	APD::SetDevice(gpu_index);
	APD::SetSeed(seed);
	std::vector<Problem> problems;
	GenerateSampleList(dense_folder, problems, max_src);
	std::cout << "There are " << problems.size() << " problems needed to be processed!" << std::endl;
	int round_num = ComputeRoundNum(problems);
	std::cout << "Round nums: " << round_num << std::endl;
	// scale of round i is 2^(round_num-1-i); rounds run while the scale is >= min_scale
	// (min_scale = 2 reproduces `i < round_num - 1`, main.cpp:450; 1 adds the full-resolution round)
	int iteration_index = 0, pass = 0;
	const path sync_dir = dense_folder / path("APD") / path(".sync");
	if (rank == 0) std::filesystem::remove_all(sync_dir);
	for (int i = 0; i < round_num; ++i) {
		const int scale = 1 << (round_num - 1 - i);
		if (scale < min_scale && !(round_num == 1 && i == 0)) break;
		for (auto& problem : problems) {
			problem.iteration = iteration_index;
			problem.scale_size = scale;
			auto& params = problem.params;
			if (i == 0) { params.state = FIRST_INIT; params.use_APD = false; }
			else {
				params.state = REFINE_INIT;
				params.use_APD = true;
				params.ransac_threshold = (float)(0.01 - i * 0.00125);
				params.rotate_time = std::min(static_cast<int>(std::pow(2, i)), 4);
				params.use_detail = true;
			}
			params.geom_consistency = false;
			params.max_iterations = iters;
			params.weak_peak_radius = 6;
			if (problem.index % world == rank) {
				GetProblemEdges(problem);   // main.cpp:480
				ProcessProblem(problem);
			}
		}
		PassBarrier(sync_dir, pass++, rank, world);
		iteration_index++;
		for (int j = 0; j < geom_passes; ++j) {
			for (auto& problem : problems) {
				problem.iteration = iteration_index;
				problem.scale_size = scale;
				auto& params = problem.params;
				params.state = REFINE_ITER;
				params.use_APD = (i != 0);
				params.ransac_threshold = (float)(0.01 - i * 0.00125);
				params.rotate_time = std::min(static_cast<int>(std::pow(2, i)), 4);
				params.geom_consistency = true;
				params.max_iterations = iters;
				params.weak_peak_radius = std::max(4 - 2 * j, 2);
				if (problem.index % world == rank) ProcessProblem(problem);
			}
			PassBarrier(sync_dir, pass++, rank, world);
			iteration_index++;
		}
		std::cout << "Round: " << i << " done\n";
	}
	APD::ReleasePooledContext();
	if (fusion && rank == 0) RunFusion(dense_folder, problems);
	std::cout << "All done\n";
	return EXIT_SUCCESS;
}
