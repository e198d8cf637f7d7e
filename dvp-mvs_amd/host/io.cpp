// io.cpp — on-disk formats of the reference's per-view pipeline, OpenCV/Boost-free.
// BinMat (APD.cpp:548-573, 630-649), ACMM dmb (APD.cpp:575-628), MVSNet camera text
// (APD.cpp:651-692), binary PLY (APD.cpp:842-882), PNM images, bilinear resize, nearest rescale.
#include "APD.h"
#include <atomic>
#include <cstring>
#include <thread>
#include <cstdio>
#include <cstdlib>

bool ReadBinMat(const path& mat_path, Mat& mat) {
	std::ifstream in(mat_path, std::ios_base::binary);
	if (!in.good()) {
		std::cerr << "Error opening file: " << mat_path << std::endl;
		return false;
	}
	int32_t version, rows, cols, type;
	in.read((char*)(&version), sizeof(int32_t));
	in.read((char*)(&rows), sizeof(int32_t));
	in.read((char*)(&cols), sizeof(int32_t));
	in.read((char*)(&type), sizeof(int32_t));
	if (!in.good() || version != 1) {
		std::cerr << "Version error: " << mat_path << std::endl;
		return false;
	}
	// header sanity: only the element types this pipeline writes, plausible sizes
	const bool known_type = type == CV_8UC1 || type == CV_8UC3 || type == CV_32SC1 || type == CV_32FC1 || type == CV_32FC3;
	if (!known_type || rows <= 0 || cols <= 0 || rows > (1 << 16) || cols > (1 << 16)) {
		std::cerr << "Header error: " << mat_path << " (" << rows << " x " << cols << ", type " << type << ")" << std::endl;
		return false;
	}
	Mat m(rows, cols, type);
	const std::streamsize want = (std::streamsize)(m.step * m.rows);
	in.read((char*)m.data, want);
	if (in.gcount() != want) {   // truncated file: leave `mat` untouched rather than hand out uninitialised rows
		std::cerr << "Short read: " << mat_path << " (" << in.gcount() << " of " << want << " bytes)" << std::endl;
		return false;
	}
	mat = m;
	return true;
}

bool WriteBinMat(const path& mat_path, const Mat& mat) {
	std::ofstream out(mat_path, std::ios_base::binary);
	if (!out.good()) {
		std::cout << "Error opening file: " << mat_path << std::endl;
		return false;
	}
	int32_t version = 1, rows = mat.rows, cols = mat.cols, type = mat.type();
	out.write((char*)&version, sizeof(int32_t));
	out.write((char*)&rows, sizeof(int32_t));
	out.write((char*)&cols, sizeof(int32_t));
	out.write((char*)&type, sizeof(int32_t));
	out.write((char*)mat.data, (std::streamsize)(mat.step * mat.rows));
	return out.good();
}

static int write_dmb(const path& p, const Mat& m, int32_t nb) {
	FILE* f = fopen(p.string().c_str(), "wb");
	if (!f) {
		std::cout << "Error opening file " << p << std::endl;
		return -1;
	}
	int32_t type = 1, h = m.rows, w = m.cols;
	fwrite(&type, sizeof(int32_t), 1, f);
	fwrite(&h, sizeof(int32_t), 1, f);
	fwrite(&w, sizeof(int32_t), 1, f);
	fwrite(&nb, sizeof(int32_t), 1, f);
	fwrite(m.data, sizeof(float), (size_t)w * h * nb, f);
	fclose(f);
	return 0;
}
int writeDepthDmb(const path& mat_path, const Mat& depth) { return write_dmb(mat_path, depth, 1); }
int writeNormalDmb(const path& mat_path, const Mat& normal) { return write_dmb(mat_path, normal, 3); }

// cams/<id>_cam.txt (MVSNet layout, written by colmap2mvsnet.py:429-441; reference reader APD.cpp:651-692):
//   "extrinsic" + 4x4 world-to-camera matrix, "intrinsic" + 3x3 K, then `depth_min interval depth_num depth_max`.
// Parsed as blocks with their keywords checked; a file that is short or mislabelled is rejected (the reference
// reads whatever the stream yields).
bool ReadCamera(const path& cam_path, Camera& cam) {
	std::ifstream in(cam_path);
	if (!in.good()) return false;
	// values are extracted straight into float, as the reference's `in >> cam.R[...]` does (APD.cpp:662-671): going
	// through double and narrowing can round a decimal twice and differ by one ulp
	auto block = [&in](const char* keyword, float* dst, int n) {
		std::string word;
		if (!(in >> word) || word != keyword) return false;
		for (int i = 0; i < n; ++i)
			if (!(in >> dst[i])) return false;
		return true;
	};
	float E[16], K[9], range[4];
	if (!block("extrinsic", E, 16) || !block("intrinsic", K, 9)) return false;
	for (float& v : range)
		if (!(in >> v)) return false;
	for (int r = 0; r < 3; ++r) {
		for (int c = 0; c < 3; ++c) {
			cam.R[3 * r + c] = E[4 * r + c];
			cam.K[3 * r + c] = K[3 * r + c];
		}
		cam.t[r] = E[4 * r + 3];
	}
	// camera centre C = -R^T t from the float32 R and t, accumulated in double (APD.cpp:673-677)
	for (int c = 0; c < 3; ++c) {
		double acc = 0.0;
		for (int r = 0; r < 3; ++r) acc += (double)cam.R[3 * r + c] * (double)cam.t[r];
		cam.c[c] = -(float)acc;
	}
	cam.depth_min = range[0];   // TAT & ETH layout (APD.cpp:680-682); interval and depth_num are not used
	cam.depth_max = range[3];
	return true;
}

// A camera file that cannot be read in full leaves no usable camera: every caller stops the job (through DvpFatal, i.e.
// through the multi-rank abort) instead of going on with an uninitialised one.
void ReadCameraOrDie(const path& cam_path, Camera& cam) {
	if (!ReadCamera(cam_path, cam)) DvpFatal("ReadCamera: " + cam_path.string() + " is missing, short or mislabelled (expected `extrinsic` 4x4, `intrinsic` 3x3, `depth_min interval depth_num depth_max`)");
}

// binary little-endian PLY: x y z float + diffuse_blue/green/red uchar per vertex (APD.cpp:842-882); PointList::color
// is already in that (B, G, R) order
bool ExportPointCloud(const path& point_cloud_path, std::vector<PointList>& pointcloud) {
	static const char* const kProperties[] = { "float x", "float y", "float z", "uchar diffuse_blue", "uchar diffuse_green", "uchar diffuse_red" };
	std::ofstream out(point_cloud_path, std::ios::binary);
	if (!out.good()) return false;
	out << "ply\nformat binary_little_endian 1.0\nelement vertex " << (int)pointcloud.size() << "\n";
	for (const char* prop : kProperties) out << "property " << prop << "\n";
	out << "end_header\n";
	constexpr size_t kRecord = 3 * sizeof(float) + 3;
	std::vector<char> records(pointcloud.size() * kRecord);
	const long long n_points = (long long)pointcloud.size();
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (long long i = 0; i < n_points; ++i) {
		const PointList& pt = pointcloud[(size_t)i];
		char* w = records.data() + (size_t)i * kRecord;
		const float xyz[3] = { pt.coord.x, pt.coord.y, pt.coord.z };
		memcpy(w, xyz, sizeof(xyz));
		w[12] = (char)(unsigned char)pt.color.x;
		w[13] = (char)(unsigned char)pt.color.y;
		w[14] = (char)(unsigned char)pt.color.z;
	}
	out.write(records.data(), (std::streamsize)records.size());
	return out.good();
}

std::string ToFormatIndex(int index) {
	std::stringstream ss;
	ss << std::setw(8) << std::setfill('0') << index;
	return ss.str();
}

// nearest-neighbour rescale with the reference's swapped scale factors (APD.cpp:1787-1788): kept
template <typename TYPE>
void RescaleMatToTargetSize(const Mat& src, Mat& dst, int tw, int th) {
	if (src.cols == tw && src.rows == th) return;
	const float scale_x = tw / static_cast<float>(src.cols);
	const float scale_y = th / static_cast<float>(src.rows);
	const Mat src_clone = src;   // shares the buffer: `out` is a fresh one, so src == dst is fine
	Mat out = Mat::zeros(th, tw, src.type());
	// (the reference divides the row index by scale_x and the column index by scale_y, APD.cpp:1787-1788: kept)
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (int r = 0; r < th; ++r) {
		const int o_r = static_cast<int>(r / scale_x);
		if (o_r < 0 || o_r >= src_clone.rows) continue;
		for (int c = 0; c < tw; ++c) {
			const int o_c = static_cast<int>(c / scale_y);
			if (o_c < 0 || o_c >= src_clone.cols) continue;
			out.at<TYPE>(r, c) = src_clone.at<TYPE>(o_r, o_c);
		}
	}
	dst = out;
}
template void RescaleMatToTargetSize<float>(const Mat&, Mat&, int, int);
template void RescaleMatToTargetSize<uint8_t>(const Mat&, Mat&, int, int);
template void RescaleMatToTargetSize<int>(const Mat&, Mat&, int, int);
template void RescaleMatToTargetSize<unsigned int>(const Mat&, Mat&, int, int);
template void RescaleMatToTargetSize<Vec3f>(const Mat&, Mat&, int, int);

// ---- PNM ---------------------------------------------------------------------------------------------
static bool read_pnm(const path& p, int want_channels, Mat& out) {
	FILE* f = fopen(p.string().c_str(), "rb");
	if (!f) return false;
	char magic[3] = { 0 };
	int w = 0, h = 0, maxv = 0;
	auto next_int = [&](int* v) {
		int c = fgetc(f);
		while (c == '#' || c == ' ' || c == '\n' || c == '\r' || c == '\t') {
			if (c == '#') while (c != '\n' && c != EOF) c = fgetc(f);
			c = fgetc(f);
		}
		int x = 0;
		while (c >= '0' && c <= '9') { x = x * 10 + (c - '0'); c = fgetc(f); }
		*v = x;
	};
	if (fread(magic, 1, 2, f) != 2) { fclose(f); return false; }
	const int ch = (magic[1] == '5') ? 1 : (magic[1] == '6' ? 3 : 0);
	if (magic[0] != 'P' || ch == 0) { fclose(f); return false; }
	next_int(&w); next_int(&h); next_int(&maxv);
	if (w <= 0 || h <= 0 || maxv != 255) { fclose(f); return false; }
	std::vector<unsigned char> buf((size_t)w * h * ch);
	if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); return false; }
	fclose(f);
	if (want_channels == 1) {
		out = Mat(h, w, CV_8UC1);
		for (size_t i = 0; i < (size_t)w * h; ++i) {
			if (ch == 1) out.data[i] = buf[i];
			else {   // RGB -> gray with OpenCV's fixed-point weights (R 0.299, G 0.587, B 0.114)
				const int r = buf[3 * i], g = buf[3 * i + 1], b = buf[3 * i + 2];
				out.data[i] = (unsigned char)((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14);
			}
		}
	} else {
		out = Mat(h, w, CV_8UC3);   // BGR like cv::imread
		for (size_t i = 0; i < (size_t)w * h; ++i) {
			if (ch == 1) out.data[3 * i] = out.data[3 * i + 1] = out.data[3 * i + 2] = buf[i];
			else { out.data[3 * i] = buf[3 * i + 2]; out.data[3 * i + 1] = buf[3 * i + 1]; out.data[3 * i + 2] = buf[3 * i]; }
		}
	}
	return true;
}
// images/<id>.jpg as the reference's converter writes it (baseline JPEG: host/jpeg.cpp); a .pgm / .ppm of
// the same stem is used when there is no .jpg (synthetic datasets are written loss-free)
static Mat read_image(const path& jpg, int channels) {
	const std::string ext = jpg.extension().string();
	if ((ext == ".jpg" || ext == ".jpeg" || ext == ".JPG" || ext == ".JPEG") && std::filesystem::exists(jpg)) {
		Mat m = DecodeJpeg(jpg, channels);
		if (!m.empty()) return m;
	}
	Mat m;
	path p = jpg;
	for (const char* ext : { ".pgm", ".ppm" }) {
		p.replace_extension(ext);
		if (std::filesystem::exists(p) && read_pnm(p, channels, m)) return m;
	}
	return Mat();
}
Mat ReadImageGray(const path& p) { return read_image(p, 1); }
// size of images/<id>.jpg (or its .pgm / .ppm stand-in) from the file header alone — a 25-Mpx JPEG takes 0.3 s to decode, its
// frame header sits in the first kilobytes.  What the driver's view -> rank assignment needs of every view on every rank.
bool ImageFileSize(const path& jpg, int* w, int* h) {
	const std::string ext = jpg.extension().string();
	if ((ext == ".jpg" || ext == ".jpeg" || ext == ".JPG" || ext == ".JPEG") && std::filesystem::exists(jpg)) {
		FILE* f = fopen(jpg.string().c_str(), "rb");
		if (f) {
			bool ok = false;
			unsigned char b[8];
			if (fread(b, 1, 2, f) == 2 && b[0] == 0xFF && b[1] == 0xD8) {
				for (;;) {   // marker segments up to the first start-of-frame (SOF0..SOF15 but DHT / JPG / DAC)
					int c = fgetc(f);
					while (c != EOF && c != 0xFF) c = fgetc(f);
					while (c == 0xFF) c = fgetc(f);
					if (c == EOF) break;
					if (c == 0xD8 || c == 0x01 || (c >= 0xD0 && c <= 0xD7)) continue;   // stand-alone markers
					if (fread(b, 1, 2, f) != 2) break;
					const int len = (b[0] << 8) | b[1];
					if (c >= 0xC0 && c <= 0xCF && c != 0xC4 && c != 0xC8 && c != 0xCC) {
						if (fread(b, 1, 5, f) == 5) { *h = (b[1] << 8) | b[2]; *w = (b[3] << 8) | b[4]; ok = *w > 0 && *h > 0; }
						break;
					}
					if (c == 0xDA || len < 2 || fseek(f, len - 2, SEEK_CUR) != 0) break;
				}
			}
			fclose(f);
			if (ok) return true;
		}
	}
	path p = jpg;
	for (const char* e : { ".pgm", ".ppm" }) {
		p.replace_extension(e);
		FILE* f = fopen(p.string().c_str(), "rb");
		if (!f) continue;
		char magic[3] = { 0 };
		int v[2] = { 0, 0 };
		bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && (magic[1] == '5' || magic[1] == '6');
		for (int i = 0; ok && i < 2; ++i) {
			int c = fgetc(f);
			while (c == '#' || c == ' ' || c == '\n' || c == '\r' || c == '\t') {
				if (c == '#') while (c != '\n' && c != EOF) c = fgetc(f);
				c = fgetc(f);
			}
			while (c >= '0' && c <= '9') { v[i] = v[i] * 10 + (c - '0'); c = fgetc(f); }
		}
		fclose(f);
		if (ok && v[0] > 0 && v[1] > 0) { *w = v[0]; *h = v[1]; return true; }
	}
	return false;
}
Mat ReadImageColor(const path& p) { return read_image(p, 3); }

// cv::resize(src, dst, Size(new_cols,new_rows), 0, 0, INTER_LINEAR) for CV_32FC1: pixel-centre
// aligned bilinear (src = (dst + 0.5) * scale - 0.5), border replicated, horizontal pass then
// vertical pass in float.  Third-party arithmetic (OpenCV >= 3.3), restated from its documentation.
Mat ResizeLinear(const Mat& src, int new_cols, int new_rows) {
	Mat dst(new_rows, new_cols, CV_32FC1);
	const double sx = (double)src.cols / new_cols, sy = (double)src.rows / new_rows;
	std::vector<int> x0(new_cols);
	std::vector<float> ax(new_cols);
	for (int dx = 0; dx < new_cols; ++dx) {
		float fx = (float)((dx + 0.5) * sx - 0.5);
		int ix = (int)std::floor(fx);
		fx -= ix;
		if (ix < 0) { ix = 0; fx = 0; }
		if (ix >= src.cols - 1) { ix = src.cols - 1; fx = 0; }
		x0[dx] = ix;
		ax[dx] = fx;
	}
#pragma omp parallel for schedule(static) num_threads(HostThreads())
	for (int dy = 0; dy < new_rows; ++dy) {
		float fy = (float)((dy + 0.5) * sy - 0.5);
		int iy = (int)std::floor(fy);
		fy -= iy;
		if (iy < 0) { iy = 0; fy = 0; }
		if (iy >= src.rows - 1) { iy = src.rows - 1; fy = 0; }
		const float* r0 = src.ptr<float>(iy);
		const float* r1 = src.ptr<float>(std::min(iy + 1, src.rows - 1));
		float* o = dst.ptr<float>(dy);
		for (int dx = 0; dx < new_cols; ++dx) {
			const int ix = x0[dx], ix1 = std::min(ix + 1, src.cols - 1);
			const float a = ax[dx];
			const float h0 = r0[ix] * (1.f - a) + r0[ix1] * a;
			const float h1 = r1[ix] * (1.f - a) + r1[ix1] * a;
			o[dx] = h0 * (1.f - fy) + h1 * fy;
		}
	}
	return dst;
}

// Host-thread budget of this process.  One rank per GPU means `world` processes share the node's cores: the driver caps
// every rank at cores / world (SetHostThreadShare) so that eight ranks on a 256-core host run 8 x 32 threads, not 8 x 32
// on top of each other's background workers.  DVP_HOST_THREADS overrides both.
static std::atomic<int> g_thread_share{0};
void SetHostThreadShare(int world) {
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	g_thread_share = world > 1 ? (int)std::max(1u, hw / (unsigned)world) : 0;
	// ... and the recycled host blocks (host/Mat.h: up to DVP_MAT_POOL_GB = 8 per process) are shared out the same way:
	// eight ranks keep 8 GB between them, not 64 next to their image and result caches
	if (world > 1) {
		matpool::Pool& p = matpool::pool();
		std::lock_guard<std::mutex> lk(p.m);
		p.cap = std::max<size_t>((size_t)1 << 30, p.cap / (size_t)world);
	}
}
static thread_local std::ostream* t_view_log = nullptr;
std::ostream& ViewLog() { return t_view_log ? *t_view_log : std::cout; }
void SetViewLog(std::ostream* buffer) { t_view_log = buffer; }
static thread_local int t_thread_cap = 0;
void SetThisThreadHostThreads(int n) { t_thread_cap = n > 0 ? n : 0; }
int HostThreads() {
	if (t_thread_cap > 0) return t_thread_cap;   // a helper thread's own (small) team: it must not take the cores of the thread that feeds the GPU
	static const int env = [] { const char* e = std::getenv("DVP_HOST_THREADS"); return e ? std::max(1, std::atoi(e)) : 0; }();
	if (env) return env;
	const unsigned hw = std::thread::hardware_concurrency();
	int n = (int)std::min(32u, std::max(1u, hw));
	const int share = g_thread_share.load();
	if (share > 0) n = std::min(n, share);
	return n;
}

static void (*g_fatal_hook)(const char*) = nullptr;
void DvpSetFatalHook(void (*hook)(const char*)) { g_fatal_hook = hook; }
void DvpFatal(const std::string& message) {
	std::cerr << message << std::endl;
	if (g_fatal_hook) g_fatal_hook(message.c_str());
	exit(EXIT_FAILURE);
}

void DvpSafeCall(int rc, dvp_ctx* ctx, const char* what, const char* file, int line) {
	if (rc != 0) {
		char buf[1024];
		snprintf(buf, sizeof(buf), "DvpSafeCall() failed at %s:%i : %s : %s", file, line, what, dvp_last_error(ctx));
		DvpFatal(buf);
	}
}
