// jpeg.cpp — baseline / extended-sequential JPEG decoder (8-bit, Huffman), so that a dense folder written
// by the reference's converter (colmap2mvsnet.py:424-469: images/%08d.jpg) is consumable as is.  Replaces
// cv::imread (APD.cpp:1057, 1842), which sits on libjpeg.
//
// Grey output is what libjpeg delivers for JCS_GRAYSCALE (OpenCV's IMREAD_GRAYSCALE on a JPEG): the luma
// plane, reconstructed with the accurate integer inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit
// constants, 2 guard bits — libjpeg's default "islow" method).  tests/test_boundary.py pins it bit for
// bit against this image's own libjpeg (through PIL, draft mode 'L').  Colour output (fusion only:
// point colours) replicates chroma samples and uses the JFIF YCbCr -> RGB equations; libjpeg's default
// "fancy" chroma interpolation is not reproduced, colours at chroma edges can differ by a few levels.
// Progressive, lossless, arithmetic-coded and 12-bit files are rejected (empty Mat + message).
#include "APD.h"
#include <cstdio>

namespace {

struct Huff {
	uint8_t bits[17] = { 0 };
	uint8_t vals[256] = { 0 };
	int mincode[17], maxcode[18], valptr[17];
	bool ok = false;
	void build() {
		int code = 0, k = 0;
		for (int l = 1; l <= 16; ++l) {
			valptr[l] = k;
			mincode[l] = code;
			code += bits[l];
			k += bits[l];
			maxcode[l] = bits[l] ? code - 1 : -1;
			code <<= 1;
		}
		maxcode[17] = 0x7fffffff;
		ok = true;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
	int blocks_w = 0, blocks_h = 0;   // in 8x8 blocks, padded to whole MCUs
	int pred = 0;
	std::vector<uint8_t> plane;       // blocks_w*8 x blocks_h*8 samples
};

struct BitReader {
	const uint8_t* p;
	const uint8_t* end;
	uint32_t acc = 0;
	int nbits = 0;
	bool hit_marker = false;
	void fill() {
		while (nbits <= 24) {
			int byte = 0;
			if (!hit_marker && p < end) {
				byte = *p++;
				if (byte == 0xFF) {
					const int nxt = p < end ? *p : 0xD9;
					if (nxt == 0) ++p;                            // stuffed zero
					else { hit_marker = true; --p; byte = 0; }    // a marker: feed zeros from here on
				}
			}
			acc |= (uint32_t)byte << (24 - nbits);
			nbits += 8;
		}
	}
	int bit() { if (nbits < 1) fill(); const int b = acc >> 31; acc <<= 1; --nbits; return b; }
	int get(int n) {   // n <= 16
		if (n == 0) return 0;
		if (nbits < n) fill();
		const int v = (int)(acc >> (32 - n));
		acc <<= n;
		nbits -= n;
		return v;
	}
	void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

int decode_symbol(BitReader& br, const Huff& h) {
	int code = 0;
	for (int l = 1; l <= 16; ++l) {
		code = (code << 1) | br.bit();
		if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
	}
	return -1;
}
inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// accurate integer IDCT (LL&M), coefficients already dequantised, natural order; out: 8x8 samples
void idct_islow(const int* in, uint8_t* out, int stride) {
	constexpr int CB = 13, P1 = 2;
	constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
	              F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
	auto descale = [](long long x, int n) { return (int)((x + (1LL << (n - 1))) >> n); };
	int ws[64];
	for (int c = 0; c < 8; ++c) {
		const int* p = in + c;
		long long z2 = p[16], z3 = p[48];
		long long z1 = (z2 + z3) * F0541;
		long long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
		z2 = p[0]; z3 = p[32];
		long long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
		const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
		tmp0 = p[56]; tmp1 = p[40]; tmp2 = p[24]; tmp3 = p[8];
		z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
		long long z4 = tmp1 + tmp3;
		const long long z5 = (z3 + z4) * F1175;
		tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
		z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
		z3 += z5; z4 += z5;
		tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
		ws[c] = descale(tmp10 + tmp3, CB - P1);      ws[56 + c] = descale(tmp10 - tmp3, CB - P1);
		ws[8 + c] = descale(tmp11 + tmp2, CB - P1);  ws[48 + c] = descale(tmp11 - tmp2, CB - P1);
		ws[16 + c] = descale(tmp12 + tmp1, CB - P1); ws[40 + c] = descale(tmp12 - tmp1, CB - P1);
		ws[24 + c] = descale(tmp13 + tmp0, CB - P1); ws[32 + c] = descale(tmp13 - tmp0, CB - P1);
	}
	for (int r = 0; r < 8; ++r) {
		const int* p = ws + 8 * r;
		long long z2 = p[2], z3 = p[6];
		long long z1 = (z2 + z3) * F0541;
		long long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
		long long tmp0 = ((long long)p[0] + p[4]) << CB, tmp1 = ((long long)p[0] - p[4]) << CB;
		const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
		tmp0 = p[7]; tmp1 = p[5]; tmp2 = p[3]; tmp3 = p[1];
		z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
		long long z4 = tmp1 + tmp3;
		const long long z5 = (z3 + z4) * F1175;
		tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
		z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
		z3 += z5; z4 += z5;
		tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
		const long long o[8] = { tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3 };
		uint8_t* q = out + (size_t)r * stride;
		for (int c = 0; c < 8; ++c) {
			const int v = descale(o[c], CB + P1 + 3) + 128;
			q[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
		}
	}
}

struct Decoder {
	std::vector<uint8_t> file;
	uint16_t qt[4][64];
	bool have_qt[4] = { false, false, false, false };
	Huff dc[4], ac[4];
	std::vector<Component> comps;
	int width = 0, height = 0, hmax = 1, vmax = 1, restart_interval = 0;
	std::string error;

	bool fail(const char* msg) { error = msg; return false; }

	bool decode(bool luma_only) {
		const uint8_t* p = file.data();
		const uint8_t* end = p + file.size();
		if (file.size() < 4 || p[0] != 0xFF || p[1] != 0xD8) return fail("not a JPEG file");
		p += 2;
		while (p + 4 <= end) {
			if (*p != 0xFF) { ++p; continue; }
			const int marker = p[1];
			p += 2;
			if (marker == 0xD8 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x01 || marker == 0xFF) { if (marker == 0xFF) --p; continue; }
			if (marker == 0xD9) break;
			const int len = (p[0] << 8) | p[1];
			if (len < 2 || p + len > end) return fail("truncated segment");
			const uint8_t* s = p + 2;
			const uint8_t* se = p + len;
			if (marker == 0xDB) {
				while (s < se) {
					const int pq = *s >> 4, tq = *s & 15;
					++s;
					if (tq > 3) return fail("bad quantisation table id");
					for (int i = 0; i < 64; ++i) {
						qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[0] << 8) | s[1]) : *s;
						s += pq ? 2 : 1;
					}
					have_qt[tq] = true;
				}
			} else if (marker == 0xC4) {
				while (s < se) {
					const int tc = *s >> 4, th = *s & 15;
					++s;
					if (th > 3 || tc > 1) return fail("bad Huffman table id");
					Huff& h = tc ? ac[th] : dc[th];
					int total = 0;
					for (int l = 1; l <= 16; ++l) { h.bits[l] = *s++; total += h.bits[l]; }
					if (total > 256 || s + total > se) return fail("bad Huffman table");
					for (int i = 0; i < total; ++i) h.vals[i] = *s++;
					h.build();
				}
			} else if (marker == 0xC0 || marker == 0xC1) {
				if (s[0] != 8) return fail("only 8-bit samples are supported");
				height = (s[1] << 8) | s[2];
				width = (s[3] << 8) | s[4];
				const int n = s[5];
				if (width <= 0 || height <= 0 || (n != 1 && n != 3)) return fail("unsupported frame (size / component count)");
				comps.assign(n, Component());
				for (int i = 0; i < n; ++i) {
					comps[i].id = s[6 + 3 * i];
					comps[i].h = s[7 + 3 * i] >> 4;
					comps[i].v = s[7 + 3 * i] & 15;
					comps[i].tq = s[8 + 3 * i];
					if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) return fail("bad sampling factors");
					if (n == 1) comps[i].h = comps[i].v = 1;   // a single-component scan is never interleaved: one block per MCU
					hmax = std::max(hmax, comps[i].h);
					vmax = std::max(vmax, comps[i].v);
				}
			} else if (marker == 0xC2 || (marker >= 0xC3 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC)) {
				return fail("progressive / lossless / arithmetic-coded JPEG is not supported (baseline only)");
			} else if (marker == 0xDD) {
				restart_interval = (s[0] << 8) | s[1];
			} else if (marker == 0xDA) {
				if (comps.empty()) return fail("scan before frame header");
				const int ns = s[0];
				if (ns != (int)comps.size()) return fail("non-interleaved multi-scan files are not supported");
				for (int i = 0; i < ns; ++i) {
					const int cid = s[1 + 2 * i];
					for (auto& c : comps)
						if (c.id == cid) { c.td = s[2 + 2 * i] >> 4; c.ta = s[2 + 2 * i] & 15; }
				}
				return scan(se, end, luma_only);
			}
			p += len;
		}
		return fail("no scan found");
	}

	bool scan(const uint8_t* data, const uint8_t* end, bool luma_only) {
		const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
		const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
		for (auto& c : comps) {
			if (!have_qt[c.tq] || !dc[c.td].ok || !ac[c.ta].ok) return fail("missing table");
			c.blocks_w = mcus_x * c.h;
			c.blocks_h = mcus_y * c.v;
			c.pred = 0;
			if (!luma_only || &c == &comps[0]) c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
		}
		BitReader br{ data, end };
		int coef[64];
		int until_restart = restart_interval;
		for (int my = 0; my < mcus_y; ++my)
			for (int mx = 0; mx < mcus_x; ++mx) {
				if (restart_interval && until_restart == 0) {
					// byte-align, skip to the RSTn marker
					const uint8_t* q = br.p;
					while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
					if (q + 1 >= end) return fail("missing restart marker");
					br.p = q + 2;
					br.reset();
					for (auto& c : comps) c.pred = 0;
					until_restart = restart_interval;
				}
				for (auto& c : comps) {
					const Huff& hd = dc[c.td];
					const Huff& ha = ac[c.ta];
					const bool want = !c.plane.empty();
					for (int by = 0; by < c.v; ++by)
						for (int bx = 0; bx < c.h; ++bx) {
							for (int i = 0; i < 64; ++i) coef[i] = 0;
							const int t = decode_symbol(br, hd);
							if (t < 0 || t > 11) return fail("corrupt DC code");
							if (t) c.pred += extend(br.get(t), t);
							coef[0] = c.pred * qt[c.tq][0];
							for (int k = 1; k < 64;) {
								const int rs = decode_symbol(br, ha);
								if (rs < 0) return fail("corrupt AC code");
								const int r = rs >> 4, sz = rs & 15;
								if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
								k += r;
								if (k > 63) return fail("corrupt AC run");
								const int z = kZigzag[k];
								coef[z] = extend(br.get(sz), sz) * qt[c.tq][z];
								++k;
							}
							if (want) {
								const int X = (mx * c.h + bx) * 8, Y = (my * c.v + by) * 8;
								idct_islow(coef, c.plane.data() + (size_t)Y * c.blocks_w * 8 + X, c.blocks_w * 8);
							}
						}
				}
				if (restart_interval) --until_restart;
			}
		return true;
	}
};

bool read_file(const path& p, std::vector<uint8_t>* out) {
	FILE* f = fopen(p.string().c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	const long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	out->resize(n > 0 ? (size_t)n : 0);
	const size_t got = n > 0 ? fread(out->data(), 1, (size_t)n, f) : 0;
	fclose(f);
	return got == out->size() && n > 0;
}

}  // namespace

// channels = 1: luma plane (libjpeg JCS_GRAYSCALE); 3: BGR
Mat DecodeJpeg(const path& file, int channels) {
	Decoder d;
	if (!read_file(file, &d.file)) return Mat();
	if (!d.decode(channels == 1)) {
		std::cerr << "DecodeJpeg: " << file << ": " << d.error << std::endl;
		return Mat();
	}
	const Component& Y = d.comps[0];
	if (channels == 1) {
		// a luma plane at reduced sampling would need interpolation; every encoder in practice gives luma full resolution
		if (Y.h != d.hmax || Y.v != d.vmax) { std::cerr << "DecodeJpeg: " << file << ": sub-sampled luma is not supported" << std::endl; return Mat(); }
		Mat g(d.height, d.width, CV_8UC1);
		for (int y = 0; y < d.height; ++y) std::memcpy(g.ptr<uint8_t>(y), Y.plane.data() + (size_t)y * Y.blocks_w * 8, (size_t)d.width);
		return g;
	}
	Mat bgr(d.height, d.width, CV_8UC3);
	for (int y = 0; y < d.height; ++y) {
		uint8_t* o = bgr.ptr<uint8_t>(y);
		for (int x = 0; x < d.width; ++x) {
			auto sample = [&](const Component& c) { return (int)c.plane[(size_t)(y * c.v / d.vmax) * c.blocks_w * 8 + (size_t)(x * c.h / d.hmax)]; };
			const int yy = sample(Y);
			int r = yy, g = yy, b = yy;
			if (d.comps.size() == 3) {
				const int cb = sample(d.comps[1]) - 128, cr = sample(d.comps[2]) - 128;
				r = (int)std::lround(yy + 1.402 * cr);
				g = (int)std::lround(yy - 0.344136 * cb - 0.714136 * cr);
				b = (int)std::lround(yy + 1.772 * cb);
			}
			o[3 * x + 0] = (uint8_t)std::min(255, std::max(0, b));
			o[3 * x + 1] = (uint8_t)std::min(255, std::max(0, g));
			o[3 * x + 2] = (uint8_t)std::min(255, std::max(0, r));
		}
	}
	return bgr;
}
