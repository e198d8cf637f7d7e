// labels.cpp — the label half of the priors: EdgeSegment(scale, image, mode = 1) of the reference
// (/root/reference/APD.cpp:120-136, 348-401, 437-499) without OpenCV.
//
// What it computes.  Low-texture regions of the image become labelled segments; the anchor search and the
// RANSAC of the weak-pixel path use the label map to extend anchors along region boundaries and to prefer
// planes that agree with the anchors' normals (APD.cu:3455-3560, 3625).  Steps:
//   1. the grey image is halved twice (bilinear), a Roberts-cross gradient is thresholded at 4: white =
//      textured, black = flat;
//   2. black pixels are grouped into 4-connected regions (Connect + Label_Update); around every region of
//      at least `weak_tex_num` pixels the one-pixel outline is traced and straight segments in it are
//      found with the progressive probabilistic Hough transform, then drawn white into the texture map —
//      this closes gaps in the outline of flat regions so that two walls meeting at a faint corner do not
//      merge;
//   3. the map is resized to the working resolution, re-thresholded, its frame is cleaned like the edge
//      map's, black pixels are grouped again: label 0 = textured, k >= 1 = a flat region, -1 = a flat region
//      of at most `weak_tex_num` pixels.
// cv::resize, cv::HoughLinesP and cv::line are third-party arithmetic (OpenCV >= 3.3, absent here): restated
// from their documented algorithms (Matas et al.'s PPHT with OpenCV's multiply-with-carry generator;
// 8-connected Bresenham line).  PARITY UNPINNED — there is no OpenCV to compare with.
#include "APD.h"
#include <vector>

namespace {

Mat to_float(const Mat& u8) {
	Mat f(u8.rows, u8.cols, CV_32FC1);
	for (int r = 0; r < u8.rows; ++r) {
		const uint8_t* s = u8.ptr<uint8_t>(r);
		float* d = f.ptr<float>(r);
		for (int c = 0; c < u8.cols; ++c) d[c] = s[c];
	}
	return f;
}
Mat to_u8(const Mat& f) {   // round to nearest, saturate
	Mat u(f.rows, f.cols, CV_8UC1);
	for (int r = 0; r < f.rows; ++r) {
		const float* s = f.ptr<float>(r);
		uint8_t* d = u.ptr<uint8_t>(r);
		for (int c = 0; c < f.cols; ++c) d[c] = (uint8_t)std::min(255L, std::max(0L, std::lrintf(s[c])));
	}
	return u;
}
Mat resize_u8(const Mat& u8, int cols, int rows) {
	if (cols == u8.cols && rows == u8.rows) return u8.clone();
	return to_u8(ResizeLinear(to_float(u8), cols, rows));
}
void binarise(Mat& m, int thr) {   // cv::threshold(THRESH_BINARY)
	for (size_t i = 0, n = (size_t)m.rows * m.cols; i < n; ++i) m.data[i] = m.data[i] > thr ? 255 : 0;
}

// Roberts cross on the interior, 50/50 on the frame (APD.cpp:120-136)
Mat RobertsCross(const Mat& src) {
	Mat dst(src.rows, src.cols, CV_8UC1);
	for (int i = 0; i < src.rows; ++i)
		for (int j = 0; j < src.cols; ++j) {
			int t1 = 50, t2 = 50;
			if (i > 0 && i < src.rows - 1 && j > 0 && j < src.cols - 1) {
				t1 = src.at<uint8_t>(i, j) - src.at<uint8_t>(i + 1, j + 1);
				t2 = src.at<uint8_t>(i + 1, j) - src.at<uint8_t>(i, j + 1);
			}
			dst.at<uint8_t>(i, j) = (uint8_t)std::sqrt((double)(t1 * t1 + t2 * t2));
		}
	return dst;
}

// 8-connected line, both end points included, clipped to the image (cv::line, thickness 1)
void draw_line(Mat& img, int x0, int y0, int x1, int y1, uint8_t value) {
	const int dx = std::abs(x1 - x0), dy = std::abs(y1 - y0);
	const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
	int err = dx - dy;
	for (;;) {
		if (x0 >= 0 && x0 < img.cols && y0 >= 0 && y0 < img.rows) img.at<uint8_t>(y0, x0) = value;
		if (x0 == x1 && y0 == y1) break;
		const int e2 = 2 * err;
		if (e2 > -dy) { err -= dy; x0 += sx; }
		if (e2 < dx) { err += dx; y0 += sy; }
	}
}

struct Segment { int x0, y0, x1, y1; };

// Progressive probabilistic Hough transform (Matas, Galambos, Kittler) in the shape of cv::HoughLinesP:
// rho = 1 px, theta = 1 degree.  Points are visited in random order; each votes for its 180 lines; when a
// bin reaches `threshold` the line is walked from the point in both directions through the mask, gaps of
// up to `max_gap` pixels are bridged, the walked points are removed (their votes withdrawn if the segment
// is long enough) and a segment of at least `min_length` is reported.
std::vector<Segment> HoughSegments(const Mat& image, int threshold, int min_length, int max_gap) {
	const int width = image.cols, height = image.rows;
	const int numangle = 180;
	const int numrho = (int)std::lround(((width + height) * 2 + 1) / 1.0);
	std::vector<int> accum((size_t)numangle * numrho, 0);
	std::vector<uint8_t> mask((size_t)width * height, 0);
	std::vector<float> trig((size_t)numangle * 2);
	for (int n = 0; n < numangle; ++n) {
		const double ang = n * (M_PI / 180.0);
		trig[2 * n] = (float)std::cos(ang);
		trig[2 * n + 1] = (float)std::sin(ang);
	}
	struct Pt { int x, y; };
	std::vector<Pt> pts;
	for (int y = 0; y < height; ++y)
		for (int x = 0; x < width; ++x)
			if (image.at<uint8_t>(y, x)) { mask[(size_t)y * width + x] = 1; pts.push_back(Pt{ x, y }); }
	uint64_t state = (uint64_t)-1;   // cv::RNG((uint64)-1), multiply-with-carry
	auto next_u32 = [&]() { state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32); return (uint32_t)state; };
	std::vector<Segment> out;
	for (int count = (int)pts.size(); count > 0; --count) {
		const int idx = (int)(next_u32() % (uint32_t)count);
		const Pt point = pts[idx];
		pts[idx] = pts[count - 1];
		if (!mask[(size_t)point.y * width + point.x]) continue;   // already swallowed by an earlier segment
		int max_val = threshold - 1, max_n = 0;
		for (int n = 0; n < numangle; ++n) {
			const int r = (int)std::lround(point.x * trig[2 * n] + point.y * trig[2 * n + 1]) + (numrho - 1) / 2;
			const int val = ++accum[(size_t)n * numrho + r];
			if (max_val < val) { max_val = val; max_n = n; }
		}
		if (max_val < threshold) continue;
		// walk along the line: the major axis advances one pixel per step, the minor one in 16.16 fixed point
		const int shift = 16;
		const float a = -trig[2 * max_n + 1], b = trig[2 * max_n];
		int x0 = point.x, y0 = point.y, dx0, dy0;
		bool xflag;
		if (std::fabs(a) > std::fabs(b)) {
			xflag = true;
			dx0 = a > 0 ? 1 : -1;
			dy0 = (int)std::lround(b * (1 << shift) / std::fabs(a));
			y0 = (y0 << shift) + (1 << (shift - 1));
		} else {
			xflag = false;
			dy0 = b > 0 ? 1 : -1;
			dx0 = (int)std::lround(a * (1 << shift) / std::fabs(b));
			x0 = (x0 << shift) + (1 << (shift - 1));
		}
		Pt line_end[2] = { point, point };
		for (int k = 0; k < 2; ++k) {
			int gap = 0, x = x0, y = y0, dx = dx0, dy = dy0;
			if (k > 0) { dx = -dx; dy = -dy; }
			for (;; x += dx, y += dy) {
				const int i1 = xflag ? x : x >> shift, j1 = xflag ? y >> shift : y;
				if (i1 < 0 || i1 >= width || j1 < 0 || j1 >= height) break;
				if (mask[(size_t)j1 * width + i1]) { gap = 0; line_end[k] = Pt{ i1, j1 }; }
				else if (++gap > max_gap) break;
			}
		}
		const bool good = std::abs(line_end[1].x - line_end[0].x) >= min_length || std::abs(line_end[1].y - line_end[0].y) >= min_length;
		for (int k = 0; k < 2; ++k) {
			int x = x0, y = y0, dx = dx0, dy = dy0;
			if (k > 0) { dx = -dx; dy = -dy; }
			for (;; x += dx, y += dy) {
				const int i1 = xflag ? x : x >> shift, j1 = xflag ? y >> shift : y;
				if (i1 < 0 || i1 >= width || j1 < 0 || j1 >= height) break;
				uint8_t& m = mask[(size_t)j1 * width + i1];
				if (m) {
					if (good)
						for (int n = 0; n < numangle; ++n) {
							const int r = (int)std::lround(i1 * trig[2 * n] + j1 * trig[2 * n + 1]) + (numrho - 1) / 2;
							--accum[(size_t)n * numrho + r];
						}
					m = 0;
				}
				if (i1 == line_end[k].x && j1 == line_end[k].y) break;
			}
		}
		if (good) out.push_back(Segment{ line_end[0].x, line_end[0].y, line_end[1].x, line_end[1].y });
	}
	return out;
}

}  // namespace

// EdgeSegment(scale, src_image, mode = 1, use_canny = false): CV_32SC1 label map at src size / 2^scale
Mat LabelSegment(const int scale, const Mat& src_image, LabelStages* stages) {
	const int robthr = 4;
	const int weak_tex_num = (int)(1.0 * src_image.rows * src_image.cols / (1024 << scale << scale));
	Mat quarter = resize_u8(src_image, src_image.cols / 2, src_image.rows / 2);
	quarter = resize_u8(quarter, quarter.cols / 2, quarter.rows / 2);
	const int unit = (int)(std::min(quarter.cols, quarter.rows) / 30.0);   // Hough threshold, minimum length and maximum gap
	Mat texture = RobertsCross(quarter);
	binarise(texture, robthr);
	if (stages) { stages->quarter = quarter.clone(); stages->texture = texture.clone(); stages->weak_tex_num = weak_tex_num; }
	{
		Mat region(texture.rows, texture.cols, CV_32S);
		std::vector<int> region_size;
		Connect(texture, region, region_size);
		Label_Update(region, region_size);
		Mat outline(texture.rows, texture.cols, CV_8UC1);
		for (size_t k = 1; k < region_size.size(); ++k) {
			if (region_size[k] < weak_tex_num) continue;
			const int id = (int)k;
			std::memset(outline.data, 0, outline.step * outline.rows);
			for (int y = 0; y < outline.rows; ++y) {
				const int* row = region.ptr<int>(y);
				for (int x = 0; x < outline.cols; ++x) {
					if (row[x] == id) continue;
					const bool touches = (x > 0 && row[x - 1] == id) || (x + 1 < outline.cols && row[x + 1] == id) ||
					                     (y > 0 && region.at<int>(y - 1, x) == id) || (y + 1 < outline.rows && region.at<int>(y + 1, x) == id);
					if (touches) outline.at<uint8_t>(y, x) = 255;
				}
			}
			for (const Segment& s : HoughSegments(outline, unit, unit, unit)) draw_line(texture, s.x0, s.y0, s.x1, s.y1, 255);
		}
	}
	if (stages) stages->texture_lines = texture.clone();
	const float factor = 1.0f / (float)(1 << scale);
	Mat map = resize_u8(texture, (int)std::round(src_image.cols * factor), (int)std::round(src_image.rows * factor));
	binarise(map, robthr);
	if (stages) stages->resized = map.clone();
	const int rows = map.rows, cols = map.cols;
	uint8_t* D = map.data;   // frame clean-up (APD.cpp:452-463)
	for (int y = 0; y < rows; y++) {
		if (D[y * cols + 1] == 0) D[y * cols] = 0;
		if (D[y * cols + cols - 2] == 0) D[y * cols + cols - 1] = 0;
	}
	for (int x = 0; x < cols; x++) {
		if (D[1 * cols + x] == 0) D[0 * cols + x] = 0;
		if (D[(rows - 2) * cols + x] == 0) D[(rows - 1) * cols + x] = 0;
	}
	if (stages) stages->cleaned = map.clone();
	Mat label(rows, cols, CV_32S);
	std::vector<int> label_size;
	Connect(map, label, label_size);
	Label_Update(label, label_size);
	for (int y = 0; y < rows; ++y) {
		int* row = label.ptr<int>(y);
		for (int x = 0; x < cols; ++x)
			if (row[x] != 0 && label_size[row[x]] <= weak_tex_num) row[x] = -1;
	}
	return label;
}
