// gather_peak.hip — calibration of the 16-byte GATHER roof of the GPU this runs on: every lane issues
// independent 16-byte loads at pseudo-random 16-byte aligned addresses of a buffer far larger than the
// Infinity Cache (4 GiB), `group` adjacent lanes sharing one 128-byte line (group 1 = every lane its own line:
// the access pattern of the weak update's anchor gathers; group 8 = a full line per 8 lanes).  Prints line
// requests per second and what that is in bytes at 64 B and at 128 B per request, so that the FETCH_SIZE /
// TCC_MISS counters of a gather kernel can be read against a measured ceiling instead of the streaming one.
//   hipcc --offload-arch=gfx950 -O3 -o gather_peak tools/gather_peak.hip ;  ./gather_peak   (profiles/r02_gather_peak.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}
typedef float f4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ void __launch_bounds__(256) gather(const f4* buf, uint32_t line_mask, int group_shift, int iters, float* out) {
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t gid = tid >> group_shift, sub = tid & ((1u << group_shift) - 1u);
	float acc = 0.0f;
	uint32_t s = mix(gid * 2654435761u + 12345u);
	for (int it = 0; it < iters; ++it) {
		f4 v[INFLIGHT];
#pragma unroll
		for (int k = 0; k < INFLIGHT; ++k) {
			s = mix(s + 0x9e3779b9u);
			const uint32_t line = s & line_mask;                  // 128-byte line index
			v[k] = buf[(size_t)line * 8 + (sub & 7u)];           // 16-byte slot inside the line
		}
#pragma unroll
		for (int k = 0; k < INFLIGHT; ++k) acc += v[k].x + v[k].w;
	}
	out[tid] = acc;
}

int main() {
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	printf("%s: %d CUs\n", p.name, cus);
	const size_t bytes = (size_t)4 << 30;
	f4* buf;
	float* out;
	if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMemset(buf, 0, bytes);
	const uint32_t line_mask = (uint32_t)(bytes / 128 - 1);
	const int blocks = cus * 8 * 4, threads = 256;     // 8 waves per SIMD worth of blocks, 4 rounds
	hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
	for (int gs : { 0, 1, 2, 3 }) {
		for (int wps : { 2, 3, 8 }) {
			const int b = cus * wps * 4;
			const int iters = 400;
			hipEvent_t e0, e1;
			hipEventCreate(&e0); hipEventCreate(&e1);
			gather<8><<<b, threads>>>(buf, line_mask, gs, 10, out);
			hipEventRecord(e0);
			gather<8><<<b, threads>>>(buf, line_mask, gs, iters, out);
			hipEventRecord(e1);
			hipEventSynchronize(e1);
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			const double lane_loads = (double)b * threads * iters * 8;
			const double lines = lane_loads / (1 << gs);
			printf("lanes/line %d, blocks/CU %2d: %.1f ms  %.1f G lane-loads/s  %.1f G line requests/s  = %.2f TB/s at 64 B, %.2f TB/s at 128 B, useful %.2f TB/s\n",
			       1 << gs, wps * 4, ms, lane_loads / ms / 1e6, lines / ms / 1e6, lines * 64 / ms / 1e9, lines * 128 / ms / 1e9, lane_loads * 16 / ms / 1e9);
		}
	}
	return 0;
}
