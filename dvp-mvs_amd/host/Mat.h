// Mat.h — minimal owning 2-D array standing in for the cv::Mat the reference's API returns
// (APD.h:105-115).  Only what the PatchMatch path and its callers use: rows/cols/type()/data/step,
// at<T>(r,c), ptr<T>(r), zeros(), clone(), empty().  Type codes are OpenCV's (they are written
// into the BinMat headers on disk, APD.cpp:630-649): CV_8UC1=0, CV_32SC1=4, CV_32FC1=5, CV_32FC3=21.
#ifndef DVP_MAT_H_
#define DVP_MAT_H_
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

enum { CV_8UC1 = 0, CV_8U = 0, CV_8UC3 = 16, CV_32SC1 = 4, CV_32S = 4, CV_32FC1 = 5, CV_32F = 5, CV_32FC3 = 21 };

struct Point { int x = 0, y = 0; Point() {} Point(int _x, int _y) : x(_x), y(_y) {} };   // cv::Point
struct Rect {                                                                           // cv::Rect
	int x = 0, y = 0, width = 0, height = 0;
	Rect() {}
	Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {}
	bool contains(const Point& p) const { return x <= p.x && p.x < x + width && y <= p.y && p.y < y + height; }
};
struct Vec3f { float v[3]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };

// Large pixel buffers are recycled: a fresh 100 MB block comes from mmap as untouched pages, and the first write — a
// memset, a file read, the engine's download — then takes 25 000 page faults (measured 20-50 ms per map at 6208x4128, on
// the critical path of every view).  A released block of >= 1 MB is kept (up to DVP_MAT_POOL_GB, default 8) and handed
// to the next request of the same size with its pages still mapped.  Contents are unspecified either way (as cv::Mat's);
// DVP_MAT_POISON=1 fills every recycled block with 0xA5 so that a reader of never-written memory shows up in the tests.
namespace matpool {
struct Pool {
	std::mutex m;
	std::multimap<size_t, uint8_t*> free_blocks;
	size_t bytes = 0, cap = 0;
	bool poison = false;
	Pool() {
		const char* e = std::getenv("DVP_MAT_POOL_GB");
		cap = (size_t)(e ? std::atoi(e) : 8) << 30;
		poison = std::getenv("DVP_MAT_POISON") != nullptr;
	}
};
inline Pool& pool() { static Pool* p = new Pool; return *p; }   // never destroyed (buffers may be released at exit)
constexpr size_t kMinPooled = (size_t)1 << 20;
inline uint8_t* acquire(size_t n) {
	if (n >= kMinPooled) {
		Pool& p = pool();
		std::unique_lock<std::mutex> lk(p.m);
		auto it = p.free_blocks.find(n);
		if (it != p.free_blocks.end()) {
			uint8_t* b = it->second;
			p.free_blocks.erase(it);
			p.bytes -= n;
			lk.unlock();
			if (p.poison) std::memset(b, 0xA5, n);
			return b;
		}
	}
	return new uint8_t[n];
}
inline void release(uint8_t* b, size_t n) {
	if (n >= kMinPooled) {
		Pool& p = pool();
		std::lock_guard<std::mutex> lk(p.m);
		if (p.bytes + n <= p.cap) { p.free_blocks.emplace(n, b); p.bytes += n; return; }
	}
	delete[] b;
}
}  // namespace matpool

class Mat {
public:
	int rows = 0, cols = 0;
	size_t step = 0;
	uint8_t* data = nullptr;
	Mat() {}
	Mat(int r, int c, int type) { create(r, c, type); }
	static size_t elem_size(int type) {
		const int depth = type & 7, ch = (type >> 3) + 1;
		static const int sz[8] = { 1, 1, 2, 2, 4, 4, 8, 2 };
		return (size_t)sz[depth] * ch;
	}
	void create(int r, int c, int type) {
		rows = r; cols = c; type_ = type;
		step = (size_t)c * elem_size(type);
		const size_t n = step * (size_t)(r > 0 ? r : 1) + 8;
		buf_ = std::shared_ptr<uint8_t[]>(matpool::acquire(n), [n](uint8_t* b) { matpool::release(b, n); });
		data = buf_.get();
	}
	static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data, 0, m.step * r); return m; }
	Mat clone() const { Mat m(rows, cols, type_); if (data) std::memcpy(m.data, data, step * rows); return m; }
	int type() const { return type_; }
	bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
	size_t total() const { return (size_t)rows * cols; }
	template <class T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + step * r + sizeof(T) * c); }
	template <class T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + step * r + sizeof(T) * c); }
	template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * r); }
	template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * r); }
private:
	int type_ = 0;
	std::shared_ptr<uint8_t[]> buf_;
};
#endif
