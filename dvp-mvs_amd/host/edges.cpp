// edges.cpp — depth-edge prior front end: EdgeSegment(mode 0, use_canny) and GetProblemEdges
// (/root/reference/APD.cpp:348-466 for mode 0; /root/reference/main.cpp:193-225).
//
// cv::Canny is third-party arithmetic (OpenCV >= 3.3, absent here): restated from OpenCV's
// documented algorithm for 8-bit input, aperture 3, L2gradient = true — Sobel 3x3 with replicated
// border, squared thresholds on dx^2+dy^2, non-maximum suppression with the tan(22.5 deg)
// fixed-point sector test, hysteresis over 8-connected candidates.  PARITY UNPINNED (no OpenCV to
// compare with).  The label path (mode 1: Roberts + HoughLinesP) lives in labels.cpp.
#include "APD.h"
#include <vector>

static Mat CannyL2(const Mat& src, double low_thresh, double high_thresh) {
	const int rows = src.rows, cols = src.cols;
	low_thresh = std::min(32767.0, low_thresh);
	high_thresh = std::min(32767.0, high_thresh);
	if (low_thresh > 0) low_thresh *= low_thresh;
	if (high_thresh > 0) high_thresh *= high_thresh;
	const int low = (int)std::floor(low_thresh), high = (int)std::floor(high_thresh);
	auto px = [&](int y, int x) -> int {
		y = y < 0 ? 0 : (y >= rows ? rows - 1 : y);
		x = x < 0 ? 0 : (x >= cols ? cols - 1 : x);
		return src.at<uint8_t>(y, x);
	};
	std::vector<short> dx((size_t)rows * cols), dy((size_t)rows * cols);
	std::vector<int> mag((size_t)(rows + 2) * (cols + 2), 0);   // zero frame
	const int ms = cols + 2;
	for (int y = 0; y < rows; ++y)
		for (int x = 0; x < cols; ++x) {
			const int gx = (px(y - 1, x + 1) + 2 * px(y, x + 1) + px(y + 1, x + 1)) - (px(y - 1, x - 1) + 2 * px(y, x - 1) + px(y + 1, x - 1));
			const int gy = (px(y + 1, x - 1) + 2 * px(y + 1, x) + px(y + 1, x + 1)) - (px(y - 1, x - 1) + 2 * px(y - 1, x) + px(y - 1, x + 1));
			dx[(size_t)y * cols + x] = (short)gx;
			dy[(size_t)y * cols + x] = (short)gy;
			mag[(size_t)(y + 1) * ms + (x + 1)] = gx * gx + gy * gy;
		}
	// 0 = candidate (passes NMS, above low), 1 = not an edge, 2 = edge
	std::vector<uint8_t> map((size_t)(rows + 2) * ms, 1);
	std::vector<int> stack;
	const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
	for (int y = 0; y < rows; ++y)
		for (int x = 0; x < cols; ++x) {
			const size_t mi = (size_t)(y + 1) * ms + (x + 1);
			const int m = mag[mi];
			if (m <= low) continue;
			const int xs = dx[(size_t)y * cols + x], ys = dy[(size_t)y * cols + x];
			const int ax = std::abs(xs);
			const long long ay = (long long)std::abs(ys) << 15;
			const long long tg22x = (long long)ax * TG22;
			bool keep;
			if (ay < tg22x) keep = m > mag[mi - 1] && m >= mag[mi + 1];
			else {
				const long long tg67x = tg22x + ((long long)ax << 16);
				if (ay > tg67x) keep = m > mag[mi - ms] && m >= mag[mi + ms];
				else {
					const int s = (xs ^ ys) < 0 ? -1 : 1;
					keep = m > mag[mi - ms - s] && m > mag[mi + ms + s];
				}
			}
			if (!keep) continue;
			if (m > high) { map[mi] = 2; stack.push_back((int)mi); }
			else map[mi] = 0;
		}
	while (!stack.empty()) {
		const int mi = stack.back();
		stack.pop_back();
		const int nb[8] = { -ms - 1, -ms, -ms + 1, -1, 1, ms - 1, ms, ms + 1 };
		for (int k = 0; k < 8; ++k) {
			const int q = mi + nb[k];
			if (map[q] == 0) { map[q] = 2; stack.push_back(q); }
		}
	}
	Mat dst(rows, cols, CV_8UC1);
	for (int y = 0; y < rows; ++y)
		for (int x = 0; x < cols; ++x) dst.at<uint8_t>(y, x) = map[(size_t)(y + 1) * ms + (x + 1)] == 2 ? 255 : 0;
	return dst;
}

// EdgeSegment(scale, src_image, mode = 0, use_canny = true) — APD.cpp:348-466
Mat EdgeSegment(const int scale, const Mat& src_image, int mode, bool use_canny) {
	if (mode == 1 && !use_canny) return LabelSegment(scale, src_image);
	if (mode != 0 || !use_canny) {
		std::cerr << "EdgeSegment: mode " << mode << " (debug image / Roberts edge map) is not available in this build\n";
		return Mat::zeros(src_image.rows, src_image.cols, CV_8UC1);
	}
	const int rows = src_image.rows, cols = src_image.cols;
	// median grey level (APD.cpp:404-427; note the loop stops at 254)
	int median_val = -1;
	float histogram[256] = { 0 };
	for (int i = 0; i < rows; ++i)
		for (int j = 0; j < cols; ++j) histogram[src_image.at<uint8_t>(i, j)]++;
	const int HalfNum = rows * cols / 2;
	int tempSum = 0;
	for (int i = 0; i < 255; i++) {
		tempSum = tempSum + (int)histogram[i];
		if (tempSum > HalfNum) { median_val = i; break; }
	}
	const float sigma = 0.67f;
	const int threshold1 = (int)((1 - sigma) * median_val);
	const int threshold2 = median_val;
	Mat dst = CannyL2(src_image, threshold1, threshold2);
	// cv::resize to the same size is the identity; cv::threshold(dst, dst, 4, 255, THRESH_BINARY)
	for (int y = 0; y < rows; ++y)
		for (int x = 0; x < cols; ++x) dst.at<uint8_t>(y, x) = dst.at<uint8_t>(y, x) > 4 ? 255 : 0;
	// border fix-ups (APD.cpp:452-463)
	uint8_t* D = dst.data;
	for (int y = 0; y < rows; y++) {
		if (D[y * cols + 1] == 0) D[y * cols] = 0;
		if (D[y * cols + cols - 2] == 0) D[y * cols + cols - 1] = 0;
	}
	for (int x = 0; x < cols; x++) {
		if (D[1 * cols + x] == 0) D[0 * cols + x] = 0;
		if (D[(rows - 2) * cols + x] == 0) D[(rows - 1) * cols + x] = 0;
	}
	return dst;
}

// GetProblemEdges (main.cpp:193-225): Canny edge map of the reference image at the current scale,
// cached as <result_folder>/edges_<scale>.dmb (+ labels_<scale>.dmb)
static int scale_level(const Problem& problem) {
	int scale = 0;
	while ((1 << scale) < problem.scale_size) scale++;
	return scale;
}
std::vector<path> ProblemEdgeOutputs(const Problem& problem) {
	const int scale = scale_level(problem);
	std::vector<path> out;
	const path edge_path = problem.result_folder / path("edges_" + std::to_string(scale) + ".dmb");
	const path label_path = problem.result_folder / path("labels_" + std::to_string(scale) + ".dmb");
	if (problem.params.use_edge && !ResultExists(edge_path)) out.push_back(edge_path);
	if (problem.params.use_label && !ResultExists(label_path)) out.push_back(label_path);
	return out;
}
void GetProblemEdges(const Problem& problem) { GetProblemEdges(problem, ProblemEdgeOutputs(problem)); }
// `outputs`: what ProblemEdgeOutputs said BEFORE the caller announced the files with ExpectResult (an announced file
// counts as existing)
void GetProblemEdges(const Problem& problem, const std::vector<path>& outputs) {
	const int scale = scale_level(problem);
	const path edge_path = problem.result_folder / path("edges_" + std::to_string(scale) + ".dmb");
	const path label_path = problem.result_folder / path("labels_" + std::to_string(scale) + ".dmb");
	bool need_edge = false, need_label = false;
	for (const path& p : outputs) {
		need_edge = need_edge || p == edge_path;
		need_label = need_label || p == label_path;
	}
	if (!need_edge && !need_label) return;
	// whatever happens below, every announced output is published (an empty map stands for "no priors")
	auto publish_empty = [&]() {
		if (need_label) PublishResult(label_path, Mat());
		if (need_edge) PublishResult(edge_path, Mat());
	};
	if (need_label) {   // from the full-size image (main.cpp:236)
		const Mat image_uint = APD::DecodedGray(problem.dense_folder / path("images") / path(ToFormatIndex(problem.ref_image_id) + ".jpg"));
		if (image_uint.empty()) { publish_empty(); return; }
		PublishResult(label_path, EdgeSegment(scale, image_uint, 1));
		need_label = false;
	}
	if (!need_edge) return;
	// the reference image at this scale: uint8 -> float -> cv::resize(INTER_LINEAR) (main.cpp:205-214) is exactly what
	// the driver's image cache holds for this view (APD.cpp load_image), decoded and resized once
	int oc = 0, orow = 0;
	const Mat scaled = APD::CachedImage(problem, problem.ref_image_id, &oc, &orow);
	if (scaled.empty()) { publish_empty(); return; }
	const int new_cols = scaled.cols, new_rows = scaled.rows;
	Mat u8(new_rows, new_cols, CV_8UC1);   // convertTo(CV_8UC1): round to nearest, saturate
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (int r = 0; r < new_rows; ++r)
		for (int c = 0; c < new_cols; ++c) u8.at<uint8_t>(r, c) = (uint8_t)std::min(255L, std::max(0L, std::lrintf(scaled.at<float>(r, c))));
	Mat edge = EdgeSegment(scale, u8, 0, true);
	PublishResult(edge_path, edge);
}
