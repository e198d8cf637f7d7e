// How long do multi-GB hipMalloc / hipFree calls take on an idle GPU and next to running kernels?  (tools/r06_run29.sh)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <atomic>
__global__ void spin(float* p, long long n, int rounds) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	float v = p[i % n];
	for (int r = 0; r < rounds; ++r) v = v * 1.0001f + 0.5f;
	p[i % n] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	(void)hipSetDevice(0);
	(void)hipFree(nullptr);
	float* w = nullptr;
	(void)hipMalloc(&w, 1 << 28);
	const long long sizes[] = { 2, 16, 48, 120 };
	for (int busy = 0; busy < 2; ++busy) {
		std::atomic<bool> stop{false};
		std::thread t;
		if (busy) t = std::thread([&]() {
			(void)hipSetDevice(0);
			hipStream_t s; (void)hipStreamCreate(&s);
			double k0 = now(); int n = 0;
			while (!stop) { hipLaunchKernelGGL(spin, dim3(65536), dim3(256), 0, s, w, (1ll << 26), 2000); (void)hipStreamSynchronize(s); ++n; }
			printf("  (busy thread: %d launches, %.2f ms each)\n", n, (now() - k0) * 1e3 / n);
		});
		if (busy) std::this_thread::sleep_for(std::chrono::milliseconds(200));
		for (long long gb : sizes) {
			void* p = nullptr;
			double t0 = now();
			hipError_t e = hipMalloc(&p, (size_t)gb << 30);
			double t1 = now();
			(void)hipMemsetAsync(p, 0, 1 << 20, 0); (void)hipDeviceSynchronize();
			double t2 = now();
			(void)hipFree(p);
			double t3 = now();
			printf("%s %3lld GB: hipMalloc %.1f ms (%s), first touch %.1f ms, hipFree %.1f ms\n", busy ? "busy" : "idle", gb, (t1 - t0) * 1e3, hipGetErrorString(e), (t2 - t1) * 1e3, (t3 - t2) * 1e3);
		}
		if (busy) { stop = true; t.join(); }
	}
	return 0;
}
