// dvp_fuse_math.hpp — the two transcendental functions of the depth-map fusion (RunFusion, /root/reference/APD.cpp:1797-1806,
// 1809-1960) as SPECIFIED functions, so that the device kernels (dvp_fuse.hip), the host path (host/fusion.cpp) and the CPU
// restatement under tests (which keeps its own statement of the same specification) give the same bits.
//
// The reference calls libm's acos / exp on the host: their low-order bits depend on the C library the binary is linked with
// (glibc's acosf / expf are < 1 ulp, not correctly rounded, and have changed between releases).  The contract here: binary32
// throughout, every step one IEEE operation (+ - * / sqrt, no contraction), < 1 ulp from the true value — the same distance
// any libm has.  Host and device compile this very text.
#ifndef DVP_FUSE_MATH_HPP_
#define DVP_FUSE_MATH_HPP_

#include "dvp_dev.hpp"

namespace dvp {

// exp: the engine's contract function (dvp_dev.hpp)
DVP_HD float fuse_expf(float x) { return dvp_expf(x); }

// acos on [-1, 1] (NaN outside): the classic three-range evaluation — a rational approximation R(z) ~ (asin(x) - x) / x^3 on
// |x| < 0.5, and acos(x) = 2 asin(sqrt((1 - x) / 2)) with a split square root above it, mirrored below -0.5.
DVP_HD float fuse_acosf(float x) {
	const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
	const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
	const uint32_t hx = f32_bits(x), ix = hx & 0x7fffffffu;
	if (ix >= 0x3f800000u) {               // |x| >= 1 (or NaN)
		if (ix == 0x3f800000u) return (hx >> 31) ? pi + 2.0f * pio2_lo : 0.0f;
		return (x - x) / (x - x);          // NaN
	}
	if (ix < 0x3f000000u) {                // |x| < 0.5
		if (ix <= 0x32800000u) return pio2_hi + pio2_lo;
		const float z = x * x;
		const float p = z * (pS0 + z * (pS1 + z * pS2));
		const float q = 1.0f + z * qS1;
		const float r = p / q;
		return pio2_hi - (x - (pio2_lo - x * r));
	}
	if (hx >> 31) {                        // x < -0.5
		const float z = (1.0f + x) * 0.5f;
		const float p = z * (pS0 + z * (pS1 + z * pS2));
		const float q = 1.0f + z * qS1;
		const float s = sqrtf(z);
		const float r = p / q;
		const float w = r * s - pio2_lo;
		return pi - 2.0f * (s + w);
	}
	const float z = (1.0f - x) * 0.5f;     // x > 0.5
	const float s = sqrtf(z);
	const float df = __builtin_bit_cast(float, f32_bits(s) & 0xfffff000u);
	const float c = (z - df * df) / (s + df);
	const float p = z * (pS0 + z * (pS1 + z * pS2));
	const float q = 1.0f + z * qS1;
	const float r = p / q;
	const float w = r * s + c;
	return 2.0f * (df + w);
}

// GetAngle (APD.cpp:1797-1806): the angle between two unit normals; the acos of a dot product rounded past 1 is NaN -> 0
DVP_HD float fuse_angle(const float* a, const float* b) {
	const float ang = fuse_acosf(a[0] * b[0] + a[1] * b[1] + a[2] * b[2]);
	return ang == ang ? ang : 0.0f;
}

}  // namespace dvp
#endif
