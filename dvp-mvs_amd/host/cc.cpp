// cc.cpp — 4-connected components of the zero pixels of a mask: what Connect + Label_Update
// (/root/reference/APD.cpp:138-346) compute for ProcessProblem's visibility-mask clean-up
// (main.cpp:323-352).  The reference builds provisional labels with a lossy union (Connect) and
// repairs them with an O(labels^2) merge (Label_Seek/Label_Update) whose loops stop at rows-1 / cols-1.
// Here: one proper union-find pass, labels renumbered in first-encounter (raster) order like the
// reference, label 0 = pixels == 255.  tests/host/test_host.cpp holds a model of the reference's
// two-step procedure and compares region sizes pixel by pixel on 300 random masks: no pixel differs
// (the up/left label propagation already joins what the truncated repair loop would miss).
#include "APD.h"
#include <numeric>

static int uf_find(std::vector<int>& p, int x) {
	while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
	return x;
}

void Connect(const Mat& img, Mat& label_mask, std::vector<int>& label_cnt) {
	const int rows = img.rows, cols = img.cols;
	std::vector<int> parent(1, 0);
	for (int y = 0; y < rows; y++) {
		for (int x = 0; x < cols; x++) {
			if (img.at<uint8_t>(y, x) == 255) { label_mask.at<int>(y, x) = 0; continue; }
			const bool left = x > 0 && img.at<uint8_t>(y, x) == 0 && img.at<uint8_t>(y, x - 1) == 0;
			const bool up = y > 0 && img.at<uint8_t>(y, x) == 0 && img.at<uint8_t>(y - 1, x) == 0;
			int l = 0;
			if (left) l = uf_find(parent, label_mask.at<int>(y, x - 1));
			if (up) {
				const int u = uf_find(parent, label_mask.at<int>(y - 1, x));
				if (l == 0) l = u;
				else if (u != l) { const int a = std::min(l, u), b = std::max(l, u); parent[b] = a; l = a; }
			}
			if (l == 0) { l = (int)parent.size(); parent.push_back(l); }
			label_mask.at<int>(y, x) = l;
		}
	}
	std::vector<int> mapping(parent.size(), 0);
	int label_num = 1;
	for (size_t i = 1; i < parent.size(); i++)
		if (uf_find(parent, (int)i) == (int)i) mapping[i] = label_num++;
	label_cnt.assign(label_num, 0);
	for (int y = 0; y < rows; y++)
		for (int x = 0; x < cols; x++) {
			const int l = label_mask.at<int>(y, x);
			const int m = (l == 0) ? 0 : mapping[uf_find(parent, l)];
			label_mask.at<int>(y, x) = m;
			label_cnt[m]++;
		}
}

// With exact components from Connect there is nothing left to merge; kept for API parity.
void Label_Update(Mat& label_mask, std::vector<int>& label_cnt) {
	std::fill(label_cnt.begin(), label_cnt.end(), 0);
	for (int y = 0; y < label_mask.rows; y++)
		for (int x = 0; x < label_mask.cols; x++) label_cnt[label_mask.at<int>(y, x)]++;
}
