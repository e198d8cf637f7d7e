// host-layer checks that need no GPU: on-disk formats, camera parser, connected components,
// rescale helpers.  Exit code 0 = all passed.
#include "../../dvp-mvs_amd/host/APD.h"
#include <cassert>
#include <cstdio>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
	path tmp = argc > 1 ? path(argv[1]) : std::filesystem::temp_directory_path();
	// BinMat round trips (APD.cpp:548-573, 630-649): header = version 1, rows, cols, cv type
	{
		Mat d(3, 5, CV_32FC1);
		for (int r = 0; r < 3; ++r) for (int c = 0; c < 5; ++c) d.at<float>(r, c) = r * 10 + c + 0.5f;
		CHECK(WriteBinMat(tmp / "t_depth.dmb", d));
		Mat e;
		CHECK(ReadBinMat(tmp / "t_depth.dmb", e));
		CHECK(e.rows == 3 && e.cols == 5 && e.type() == CV_32FC1);
		CHECK(std::memcmp(d.data, e.data, 3 * 5 * 4) == 0);
		FILE* f = fopen((tmp / "t_depth.dmb").string().c_str(), "rb");
		int32_t h[4];
		CHECK(fread(h, 4, 4, f) == 4);
		fclose(f);
		CHECK(h[0] == 1 && h[1] == 3 && h[2] == 5 && h[3] == 5);
		CHECK(std::filesystem::file_size(tmp / "t_depth.dmb") == 16 + 3 * 5 * 4);
		Mat n(2, 2, CV_32FC3), w(2, 2, CV_8UC1), s(2, 2, CV_32SC1);
		CHECK(n.step == 24 && w.step == 2 && s.step == 8 && n.type() == 21 && s.type() == 4);
		CHECK(!ReadBinMat(tmp / "does_not_exist.dmb", e));
		CHECK(writeDepthDmb(tmp / "t_acmm.dmb", d) == 0);
		CHECK(std::filesystem::file_size(tmp / "t_acmm.dmb") == 16 + 3 * 5 * 4);
	}
	// camera text (APD.cpp:651-692): c = -R^T t, depth_min interval depth_num depth_max
	{
		std::ofstream f(tmp / "t_cam.txt");
		f << "extrinsic\n0 -1 0 1\n1 0 0 2\n0 0 1 3\n0 0 0 1\n\nintrinsic\n500 0 320\n0 510 240\n0 0 1\n\n2.5 0.02 192 6.5\n";
		f.close();
		Camera cam;
		CHECK(ReadCamera(tmp / "t_cam.txt", cam));
		CHECK(cam.R[1] == -1 && cam.R[3] == 1 && cam.t[2] == 3 && cam.K[0] == 500 && cam.K[4] == 510 && cam.K[5] == 240);
		CHECK(cam.depth_min == 2.5f && cam.depth_max == 6.5f);
		// c_j = -(R[0+j] t0 + R[3+j] t1 + R[6+j] t2)
		CHECK(cam.c[0] == -2.0f && cam.c[1] == 1.0f && cam.c[2] == -3.0f);
		CHECK(!ReadCamera(tmp / "missing_cam.txt", cam));
	}
	CHECK(ToFormatIndex(42) == "00000042");
	// connected components of zero pixels, label 0 = 255-pixels (APD.cpp:233-346)
	{
		const char* rows[5] = { "0.0..", "0.0.0", "...00", "0....", "00.0." };   // '0' = zero pixel, '.' = 255
		Mat m(5, 5, CV_8UC1), lab(5, 5, CV_32S);
		for (int r = 0; r < 5; ++r) for (int c = 0; c < 5; ++c) m.at<uint8_t>(r, c) = rows[r][c] == '0' ? 0 : 255;
		std::vector<int> cnt;
		Connect(m, lab, cnt);
		Label_Update(lab, cnt);
		CHECK(cnt.size() == 6);   // background + 5 components: {0,0;1,0} {0,2;1,2} {1,4;2,3;2,4} {3,0;4,0;4,1} {4,3}
		CHECK(lab.at<int>(0, 0) == lab.at<int>(1, 0) && lab.at<int>(0, 0) != lab.at<int>(0, 2));
		CHECK(lab.at<int>(1, 4) == lab.at<int>(2, 3) && lab.at<int>(2, 3) == lab.at<int>(2, 4));
		CHECK(lab.at<int>(3, 0) == lab.at<int>(4, 1));
		CHECK(cnt[lab.at<int>(1, 4)] == 3 && cnt[lab.at<int>(4, 3)] == 1 && lab.at<int>(0, 1) == 0);
		int total = 0;
		for (int v : cnt) total += v;
		CHECK(total == 25);
	}
	// rescale helpers
	{
		Mat a(4, 6, CV_32FC1);
		for (int r = 0; r < 4; ++r) for (int c = 0; c < 6; ++c) a.at<float>(r, c) = (float)(r * 6 + c);
		Mat same = ResizeLinear(a, 6, 4);
		CHECK(std::memcmp(same.data, a.data, 4 * 6 * 4) == 0);
		Mat half = ResizeLinear(a, 3, 2);
		CHECK(half.rows == 2 && half.cols == 3);
		CHECK(std::fabs(half.at<float>(0, 0) - 3.5f) < 1e-5f);   // mean of the 2x2 block {0,1,6,7}
		Mat up;
		RescaleMatToTargetSize<float>(a, up, 12, 8);
		CHECK(up.rows == 8 && up.cols == 12 && up.at<float>(0, 0) == 0.0f && up.at<float>(7, 11) == a.at<float>(3, 5));
	}
	// PLY header (APD.cpp:842-882)
	{
		std::vector<PointList> pc(2);
		pc[0].coord = float3{1, 2, 3}; pc[0].color = float3{10, 20, 30};
		pc[1].coord = float3{4, 5, 6}; pc[1].color = float3{40, 50, 60};
		CHECK(ExportPointCloud(tmp / "t.ply", pc));
		std::ifstream f(tmp / "t.ply", std::ios::binary);
		std::string l;
		std::getline(f, l); CHECK(l == "ply");
		std::getline(f, l); CHECK(l == "format binary_little_endian 1.0");
		std::getline(f, l); CHECK(l == "element vertex 2");
		int lines = 3;
		while (std::getline(f, l)) { lines++; if (l == "end_header") break; }
		CHECK(lines == 10);
		std::vector<char> rest((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
		CHECK(rest.size() == 2 * 15);
	}
	// Canny edge front end (EdgeSegment mode 0): a vertical step must give one thin vertical edge
	{
		Mat img(40, 60, CV_8UC1);
		for (int r = 0; r < 40; ++r) for (int c = 0; c < 60; ++c) img.at<uint8_t>(r, c) = c < 30 ? 60 : 180;
		Mat e = EdgeSegment(0, img, 0, true);
		CHECK(e.rows == 40 && e.cols == 60);
		int on_step = 0, elsewhere = 0;
		for (int r = 2; r < 38; ++r)
			for (int c = 0; c < 60; ++c)
				if (e.at<uint8_t>(r, c)) { if (c == 29 || c == 30) on_step++; else elsewhere++; }
		CHECK(on_step == 36 && elsewhere == 0);
		Mat flat = Mat::zeros(20, 20, CV_8UC1);
		Mat e2 = EdgeSegment(0, flat, 0, true);
		int any = 0;
		for (int i = 0; i < 400; ++i) any += e2.data[i];
		CHECK(any == 0);
	}
	static_assert(sizeof(Camera) == 112 && sizeof(PatchMatchParams) == 76, "POD layouts");
	printf("host tests ok\n");
	return 0;
}
