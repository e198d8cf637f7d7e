// valu_peak.hip — calibration of the non-packed fp32 VALU issue rate of the GPU this runs on: every lane
// runs `iters` rounds of 16 independent v_fma_f32 (no memory traffic), at 1..8 waves per SIMD.  Prints wave64
// VALU instructions per second; bench.py's "valu" roofline uses the figure measured on MI355X
// (profiles/r02_valu_peak.txt).   hipcc --offload-arch=gfx950 -O3 -o valu_peak tools/valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void fma_chain(float* out, int iters, float a, float b) {
	float x[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) x[i] = (float)(threadIdx.x + i) * 1e-3f;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
	}
	float s = 0.0f;
#pragma unroll
	for (int i = 0; i < 16; ++i) s += x[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f2v __attribute__((ext_vector_type(2)));
// the same with v_pk_fma_f32 (two fp32 FMAs per lane and instruction): 8 independent packed chains
__global__ void pk_fma_chain(float* out, int iters, float a, float b) {
	f2v x[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) { x[i].x = (float)(threadIdx.x + i) * 1e-3f; x[i].y = (float)(threadIdx.x + i) * 2e-3f; }
	const f2v av = { a, a }, bv = { b, b };
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
	}
	float s = 0.0f;
#pragma unroll
	for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	printf("%s: %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
	float* out;
	hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * 4 * sizeof(float));
	const int iters = 200000;
	for (int wps : { 1, 2, 3, 4, 6, 8 }) {   // waves per SIMD
		const int blocks = cus * wps, threads = 256;   // one 4-wave block per (CU, wave slot): one wave per SIMD each (3, 6: as the dispatcher places them)
		hipEvent_t a, b;
		hipEventCreate(&a); hipEventCreate(&b);
		fma_chain<<<blocks, threads>>>(out, 1000, 1.0001f, 0.5f);
		hipEventRecord(a);
		fma_chain<<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms = 0;
		hipEventElapsedTime(&ms, a, b);
		const double insts = (double)blocks * 4 * (double)iters * 16;   // wave-level v_fma_f32 (loop overhead: ~2 SALU per round, not counted)
		printf("waves/SIMD %d: %.1f ms, %.1f G wave-instr/s = %.2f cycles per instruction per SIMD at 2.4 GHz\n", wps, ms, insts / ms / 1e6,
		       (double)cus * 4 * 2.4e9 / (insts / (ms * 1e-3)));
	}
	for (int wps : { 1, 2, 3, 4, 8 }) {
		const int blocks = cus * wps, threads = 256;
		hipEvent_t a, b;
		hipEventCreate(&a); hipEventCreate(&b);
		pk_fma_chain<<<blocks, threads>>>(out, 1000, 1.0001f, 0.5f);
		hipEventRecord(a);
		pk_fma_chain<<<blocks, threads>>>(out, iters, 1.0001f, 0.5f);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms = 0;
		hipEventElapsedTime(&ms, a, b);
		const double insts = (double)blocks * 4 * (double)iters * 8;
		printf("packed, waves/SIMD %d: %.1f ms, %.1f G wave v_pk_fma_f32/s = %.1f G fp32 FMA-equivalents/s\n", wps, ms, insts / ms / 1e6, 2 * insts / ms / 1e6);
	}
	return 0;
}
