// prior.cpp — monocular-depth plane prior of a FIRST_INIT pass (/root/reference/APD.cpp:1210-1424).
//
// Inputs, per reference view:  dep/<id>.dmb   (BinMat f32, a relative depth map from Depth-Anything-V2;
//                                              the network itself is external and never part of this repo)
//                              sfm/<id>.txt   (one sparse SfM point per line: x2d y2d X Y Z r g b)
//                              cams/<id>_cam.txt
// What the reference does with them and this file restates:
//   1. dep <- 255 - dep                                                        (APD.cpp:1221-1225)
//   2. for every sparse point that projects inside the map: rate = dep(proj) / projected depth
//                                                                              (APD.cpp:1254-1268)
//   3. Delaunay-triangulate the 2-D positions; fill a rate map by barycentric interpolation inside
//      every triangle, `rates[n/2]` (the middle element, not a median) elsewhere (APD.cpp:1276-1350)
//   4. dep <- dep / rate map  (now metric), rescale to the working size        (APD.cpp:1352-1363)
//   5. normals from forward differences of the back-projected depth, flipped toward the camera and
//      rotated to the world frame; plane = (world normal, depth)               (APD.cpp:1365-1422)
// The debug pictures the reference writes on the way (COLMAP_*.jpg, Tri_*.jpg, depth_anything_*.jpg,
// normal_COLMAP.jpg) are not produced.
//
// Third-party arithmetic: the reference triangulates with OpenCV's cv::Subdiv2D (OpenCV >= 3.3,
// README.md:28; not vendored).  Its published behaviour is the Delaunay triangulation of the inserted
// points plus three virtual vertices of an enclosing triangle A=(x0+3m, y0), B=(x0, y0+3m),
// C=(x0-3m, y0-3m), m = max(width, height) of the bounding rectangle; duplicate insertions are
// ignored.  The triangulation is unique for points in general position, so an incremental
// Bowyer-Watson construction over the same vertex set yields the same triangle *set*; the order of
// the list (which decides the winner where rasterised triangles overlap by a pixel) is not pinned.
#include "APD.h"
#include <array>
#include <set>

namespace {

struct DTri { int a, b, c; double cx, cy, r2; bool alive; };

// > 0 when p lies inside the circumcircle of the counter-clockwise triangle (a, b, c)
static long double in_circle(const double* a, const double* b, const double* c, const double* p) {
	const long double ax = (long double)a[0] - p[0], ay = (long double)a[1] - p[1];
	const long double bx = (long double)b[0] - p[0], by = (long double)b[1] - p[1];
	const long double cx = (long double)c[0] - p[0], cy = (long double)c[1] - p[1];
	const long double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
	return ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx);
}

static double orient(const double* a, const double* b, const double* c) {
	return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}

struct Delaunay {
	std::vector<std::array<double, 2>> pts;   // 0..2 = virtual outer vertices
	std::vector<DTri> tris;
	size_t dead = 0;

	void add_tri(int a, int b, int c) {
		if (orient(pts[a].data(), pts[b].data(), pts[c].data()) < 0) std::swap(b, c);
		DTri t{ a, b, c, 0, 0, 0, true };
		const double* A = pts[a].data(); const double* B = pts[b].data(); const double* C = pts[c].data();
		const double bx = B[0] - A[0], by = B[1] - A[1], cx = C[0] - A[0], cy = C[1] - A[1];
		const double d = 2.0 * (bx * cy - by * cx);
		if (std::fabs(d) > 1e-300) {
			const double ux = (cy * (bx * bx + by * by) - by * (cx * cx + cy * cy)) / d;
			const double uy = (bx * (cx * cx + cy * cy) - cx * (bx * bx + by * by)) / d;
			t.cx = A[0] + ux; t.cy = A[1] + uy; t.r2 = ux * ux + uy * uy;
		} else {
			t.r2 = -1.0;   // degenerate: always take the exact predicate
		}
		tris.push_back(t);
	}

	bool circum_contains(const DTri& t, const double* p) const {
		if (t.r2 >= 0.0) {   // cheap filter with a relative safety margin, exact-ish predicate in the band
			const double dx = p[0] - t.cx, dy = p[1] - t.cy, d2 = dx * dx + dy * dy;
			if (d2 > t.r2 * (1.0 + 1e-7)) return false;
			if (d2 < t.r2 * (1.0 - 1e-7)) return true;
		}
		return in_circle(pts[t.a].data(), pts[t.b].data(), pts[t.c].data(), p) > 0.0L;
	}

	void insert(double x, double y) {
		const int pi = (int)pts.size();
		pts.push_back({ x, y });
		const double p[2] = { x, y };
		std::vector<std::pair<int, int>> edges;
		for (auto& t : tris) {
			if (!t.alive || !circum_contains(t, p)) continue;
			t.alive = false;
			dead++;
			edges.emplace_back(t.a, t.b);
			edges.emplace_back(t.b, t.c);
			edges.emplace_back(t.c, t.a);
		}
		// cavity boundary = edges that belong to exactly one removed triangle
		std::vector<std::pair<int, int>> key(edges.size());
		for (size_t i = 0; i < edges.size(); ++i) key[i] = std::minmax(edges[i].first, edges[i].second);
		std::vector<size_t> ord(edges.size());
		for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
		std::sort(ord.begin(), ord.end(), [&](size_t l, size_t r) { return key[l] < key[r]; });
		for (size_t i = 0; i < ord.size();) {
			size_t j = i + 1;
			while (j < ord.size() && key[ord[j]] == key[ord[i]]) ++j;
			if (j - i == 1) add_tri(edges[ord[i]].first, edges[ord[i]].second, pi);
			i = j;
		}
		if (dead > 4096 && dead * 2 > tris.size()) {   // compact
			std::vector<DTri> keep;
			keep.reserve(tris.size() - dead);
			for (const auto& t : tris) if (t.alive) keep.push_back(t);
			tris.swap(keep);
			dead = 0;
		}
	}
};

}  // namespace

// APD.cpp:24-49
static double triangleArea(const double* A, const double* B, const double* C) {
	return 0.5 * std::abs(A[0] * (B[1] - C[1]) + B[0] * (C[1] - A[1]) + C[0] * (A[1] - B[1]));
}

double calculateZ(const double A[3], const double B[3], const double C[3], double X, double Y) {
	const double P[3] = { X, Y, 0 };
	const double areaABC = triangleArea(A, B, C);
	const double u = triangleArea(P, B, C) / areaABC;
	const double v = triangleArea(P, C, A) / areaABC;
	const double w = triangleArea(P, A, B) / areaABC;
	return u * A[2] + v * B[2] + w * C[2];
}

// APD.cpp:51-80.  Points outside [0,cols)x[0,rows) are not inserted; triangle corners are the
// integer-truncated vertex positions; the three rates are looked up by exact coordinate match.
std::vector<Triangle> DelaunayTriangulation(int cols, int rows, const Rect boundRC, std::vector<float2> xy_temps, std::vector<float> rates) {
	Delaunay dt;
	const double big = 3.0 * std::max(boundRC.width, boundRC.height);
	dt.pts.push_back({ boundRC.x + big, (double)boundRC.y });
	dt.pts.push_back({ (double)boundRC.x, boundRC.y + big });
	dt.pts.push_back({ boundRC.x - big, boundRC.y - big });
	dt.add_tri(0, 1, 2);
	std::vector<int> owner;                     // dt.pts index - 3 -> index into xy_temps (last duplicate wins, as the reference's scan does)
	std::set<std::pair<float, float>> seen;
	for (size_t i = 0; i < xy_temps.size(); ++i) {
		const float2 q = xy_temps[i];
		if (!(q.x >= 0 && q.x < cols && q.y >= 0 && q.y < rows)) continue;
		if (!seen.insert({ q.x, q.y }).second) {
			for (size_t k = 0; k < owner.size(); ++k)
				if (xy_temps[owner[k]].x == q.x && xy_temps[owner[k]].y == q.y) owner[k] = (int)i;
			continue;
		}
		owner.push_back((int)i);
		dt.insert(q.x, q.y);
	}
	std::vector<Triangle> results;
	auto corner = [&](int v, Point* p, float* rate) {
		p->x = (int)(float)dt.pts[v][0];
		p->y = (int)(float)dt.pts[v][1];
		*rate = v >= 3 ? rates[owner[v - 3]] : 0.0f;
	};
	for (const auto& t : dt.tris) {
		if (!t.alive) continue;
		Point p1, p2, p3;
		float r1, r2, r3;
		corner(t.a, &p1, &r1);
		corner(t.b, &p2, &r2);
		corner(t.c, &p3, &r3);
		Triangle tri(p1, p2, p3);
		tri.rate1 = r1; tri.rate2 = r2; tri.rate3 = r3;
		results.push_back(tri);
	}
	return results;
}

static float3 Get3DPoint(const Camera& camera, int x, int y, const float depth) {   // APD.cpp:527-534
	float3 X;
	X.x = depth * (x - camera.K[2]) / camera.K[0];
	X.y = depth * (y - camera.K[5]) / camera.K[4];
	X.z = depth;
	return X;
}

void ProjectCamera(const float3 PointX, const Camera camera, float2& point, float& depth) {   // APD.cpp:536-546
	float3 tmp;
	tmp.x = camera.R[0] * PointX.x + camera.R[1] * PointX.y + camera.R[2] * PointX.z + camera.t[0];
	tmp.y = camera.R[3] * PointX.x + camera.R[4] * PointX.y + camera.R[5] * PointX.z + camera.t[1];
	tmp.z = camera.R[6] * PointX.x + camera.R[7] * PointX.y + camera.R[8] * PointX.z + camera.t[2];
	depth = camera.K[6] * tmp.x + camera.K[7] * tmp.y + camera.K[8] * tmp.z;
	point.x = (camera.K[0] * tmp.x + camera.K[1] * tmp.y + camera.K[2] * tmp.z) / depth;
	point.y = (camera.K[3] * tmp.x + camera.K[4] * tmp.y + camera.K[5] * tmp.z) / depth;
}

// Steps 1-4: relative map + sparse points -> metric depth at the map's own resolution.
// `cam` is the camera as read from cams/<id>_cam.txt (unscaled, APD.cpp:1258-1260).
bool MetricDepthFromPrior(Mat& dep, const std::vector<float2>& xy, const std::vector<float3>& xyz, const Camera& cam) {
	if (dep.empty()) return false;
	for (int y = 0; y < dep.rows; y++)
		for (int x = 0; x < dep.cols; x++) dep.at<float>(y, x) = 255 - dep.at<float>(y, x);
	std::vector<float2> xy_temps;
	std::vector<float> rates;
	for (size_t i = 0; i < xy.size(); i++) {
		float2 point;
		float proj_depth;
		ProjectCamera(xyz[i], cam, point, proj_depth);
		const int ix = int(point.x + 0.5f), iy = int(point.y + 0.5f);
		if (ix > 0 && ix < dep.cols && iy > 0 && iy < dep.rows) {
			rates.push_back(dep.at<float>(iy, ix) / proj_depth);
			xy_temps.push_back(xy[i]);
		}
	}
	if (rates.empty()) return false;   // the reference indexes rates[0] of an empty vector here
	const float middle_rate = rates[rates.size() / 2];
	Mat all_rate_map(dep.rows, dep.cols, CV_32FC1);
	for (int y = 0; y < dep.rows; y++)
		for (int x = 0; x < dep.cols; x++) all_rate_map.at<float>(y, x) = middle_rate;
	const Rect imageRC(0, 0, dep.cols, dep.rows);
	const auto triangles = DelaunayTriangulation(dep.cols, dep.rows, imageRC, xy_temps, rates);
	for (const auto& triangle : triangles) {
		if (!(imageRC.contains(triangle.pt1) && imageRC.contains(triangle.pt2) && imageRC.contains(triangle.pt3))) continue;
		const float L01 = sqrt(pow(triangle.pt1.x - triangle.pt2.x, 2) + pow(triangle.pt1.y - triangle.pt2.y, 2));
		const float L02 = sqrt(pow(triangle.pt1.x - triangle.pt3.x, 2) + pow(triangle.pt1.y - triangle.pt3.y, 2));
		const float L12 = sqrt(pow(triangle.pt2.x - triangle.pt3.x, 2) + pow(triangle.pt2.y - triangle.pt3.y, 2));
		const float max_edge_length = std::max(L01, std::max(L02, L12));
		if (!(max_edge_length > 0.0f)) continue;   // three corners on one pixel: step would be inf
		const float step = 1.0 / max_edge_length;
		const double A[3] = { (double)triangle.pt1.x, (double)triangle.pt1.y, triangle.rate1 };
		const double B[3] = { (double)triangle.pt2.x, (double)triangle.pt2.y, triangle.rate2 };
		const double C[3] = { (double)triangle.pt3.x, (double)triangle.pt3.y, triangle.rate3 };
		if (!(triangleArea(A, B, C) > 0.0)) continue;   // collinear integer corners: 0/0 in calculateZ
		// barycentric sweep of APD.cpp:1333-1347 (float loop counters, truncating pixel casts)
		for (float p = 0; p < 1.0; p += step) {
			for (float q = 0; q < 1.0 - p; q += step) {
				const int x = p * triangle.pt1.x + q * triangle.pt2.x + (1.0 - p - q) * triangle.pt3.x;
				const int y = p * triangle.pt1.y + q * triangle.pt2.y + (1.0 - p - q) * triangle.pt3.y;
				all_rate_map.at<float>(y, x) = (float)calculateZ(A, B, C, x, y);
			}
		}
	}
	for (int y = 0; y < dep.rows; y++)
		for (int x = 0; x < dep.cols; x++) dep.at<float>(y, x) /= all_rate_map.at<float>(y, x);
	return true;
}

// Step 5: planes (world normal, depth) from a metric depth map at the working size.
// Border pixels keep a zero normal (APD.cpp:1367-1368 loops over the interior only).
void PlanesFromDepth(const Mat& dep, const Camera& cam, float4* planes) {
	const int width = dep.cols, height = dep.rows;
	for (int i = 0; i < width * height; ++i) planes[i] = float4{ 0, 0, 0, dep.at<float>(i / width, i % width) };
	for (int y = 1; y < height - 1; ++y) {
		for (int x = 1; x < width - 1; ++x) {
			const float3 X = Get3DPoint(cam, x, y, dep.at<float>(y, x));
			const float3 X_dx = Get3DPoint(cam, x + 1, y, dep.at<float>(y, x + 1));
			const float3 X_dy = Get3DPoint(cam, x, y + 1, dep.at<float>(y + 1, x));
			const float ax = X_dx.x - X.x, ay = X_dx.y - X.y, az = X_dx.z - X.z;
			const float bx = X_dy.x - X.x, by = X_dy.y - X.y, bz = X_dy.z - X.z;
			float n[3] = { ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx };
			// cv::normalize(Vec3f): L2 norm accumulated in double, zero vector stays zero
			const double len = std::sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
			const double inv = len != 0.0 ? 1.0 / len : 0.0;
			n[0] = (float)(n[0] * inv); n[1] = (float)(n[1] * inv); n[2] = (float)(n[2] * inv);
			const float norm = sqrt(X.x * X.x + X.y * X.y + X.z * X.z);
			const float vx = X.x / norm, vy = X.y / norm, vz = X.z / norm;
			if (n[0] * vx + n[1] * vy + n[2] * vz > 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
			float4& pl = planes[y * width + x];   // world = R^T n
			pl.x = cam.R[0] * n[0] + cam.R[3] * n[1] + cam.R[6] * n[2];
			pl.y = cam.R[1] * n[0] + cam.R[4] * n[1] + cam.R[7] * n[2];
			pl.z = cam.R[2] * n[0] + cam.R[5] * n[1] + cam.R[8] * n[2];
		}
	}
}

// The whole block for one problem.  Returns false (planes untouched) when dep/ or sfm/ inputs are
// missing or unusable; the caller then leaves the planes at zero and RandomInitialization draws
// random ones (APD.cu:1289-1291) — the reference has no such fallback and reads an empty Mat.
bool BuildPlanePrior(const Problem& problem, const Camera& scaled_ref_camera, int width, int height, float4* planes) {
	const path dep_path = problem.dense_folder / path("dep") / path(ToFormatIndex(problem.ref_image_id) + ".dmb");
	const path sfm_path = problem.dense_folder / path("sfm") / path(ToFormatIndex(problem.ref_image_id) + ".txt");
	if (!std::filesystem::exists(dep_path) || !std::filesystem::exists(sfm_path)) return false;
	Mat dep;
	if (!ReadBinMat(dep_path, dep) || dep.empty() || dep.type() != CV_32FC1) return false;
	std::vector<float2> xy;
	std::vector<float3> xyz;
	{
		std::ifstream file(sfm_path);
		std::string line;
		while (std::getline(file, line)) {   // APD.cpp:1241-1250
			std::istringstream iss(line);
			float x_2d = 0, y_2d = 0, x_3d = 0, y_3d = 0, z_3d = 0;
			int r, g, b;
			iss >> x_2d >> y_2d >> x_3d >> y_3d >> z_3d >> r >> g >> b;
			xy.push_back(float2{ x_2d, y_2d });
			xyz.push_back(float3{ x_3d, y_3d, z_3d });
		}
	}
	Camera cam;
	if (!ReadCamera(problem.dense_folder / path("cams") / path(ToFormatIndex(problem.ref_image_id) + "_cam.txt"), cam)) return false;
	if (!MetricDepthFromPrior(dep, xy, xyz, cam)) return false;
	if (dep.cols != width || dep.rows != height) RescaleMatToTargetSize<float>(dep, dep, width, height);
	PlanesFromDepth(dep, scaled_ref_camera, planes);
	return true;
}
