// dvp_weak.hpp — per-pixel bodies of the weak-pixel path: prior preparation (GenEdgeInform),
// anchor search (FindNearestStrongPoint, GenNeighbours, NeigbourUpdate), RANSAC fit plane and the
// deformable-NCC weak update.  These touch only WEAK pixels (1-30 % of a view).
#ifndef DVP_WEAK_HPP_
#define DVP_WEAK_HPP_

#include "dvp_strong.hpp"

namespace dvp {

#include "dvp_sector5.inc"

// ---- GenEdgeInform (APD.cu:3731-3890) ----------------------------------------------------------
// The reference bins the 120 offsets of the 11x11 window into 30-degree sectors with fp64 atan2
// per offset, bubble-sorts each sector and then the 12 winners (24-byte structs in scratch).
// Here the sector of an offset comes from a 121-entry table and each sector keeps a running
// arg-max; the stable descending sort of the 12 winners is kept.
// Nearest edge pixel in each of the 8 directions (APD.cu:3799-3824).  The reference marches a ray
// per pixel and direction (up to max(W,H) steps each: O(L*(W+H)) reads).  Here one thread owns
// one image line of one direction and walks it once AGAINST the direction, carrying the last edge
// pixel seen: O(L) reads per direction, same result.
DVP_HD int edge_ray_lines(int W, int H, int k) { return k < 2 ? W : (k < 4 ? H : W + H - 1); }
// what = 0: nearest edge pixel -> edge_neigh;  what = 1: nearest pixel with label -1 -> label_stop
// (the walks of the label-boundary search stop there, APD.cu:3861-3873)
DVP_HD void edge_ray_line(const Dev& d, int k, int line, int what = 0) {
	const int W = d.width, H = d.height;
	const int dxs[8] = { 0, 0, -1, 1, -1, 1, -1, 1 };
	const int dys[8] = { -1, 1, 0, 0, -1, 1, 1, -1 };
	const int dx = dxs[k], dy = dys[k];
	if (line >= edge_ray_lines(W, H, k)) return;
	// first pixel of the line in walking order (-d): the pixel whose +d neighbour is outside
	int x, y;
	if (dx == 0) { x = line; y = dy < 0 ? 0 : H - 1; }
	else if (dy == 0) { x = dx < 0 ? 0 : W - 1; y = line; }
	else {
		const int xb = dx < 0 ? 0 : W - 1, yb = dy < 0 ? 0 : H - 1;
		if (line < H) { x = xb; y = line; }
		else { const int t = line - H; x = (xb == 0) ? t + 1 : t; y = yb; }
	}
	s2 last = mks2(-1, -1);
	while (x >= 0 && x < W && y >= 0 && y < H) {
		const int c = x + y * W;
		if (what == 0) {
			d.edge_neigh[(size_t)c * 8 + k] = last;
			if (d.edge[c]) last = mks2(x, y);
		} else {
			d.label_stop[(size_t)c * 8 + k] = last;
			if (d.label[c] == -1) last = mks2(x, y);
		}
		x -= dx;
		y -= dy;
	}
}

// Visibility-prior tap candidates of (pixel, source view v) (APD.cu:3746-3794): the window offsets whose
// pixel sees view v are binned into twelve 30-degree sectors (at most 20 per sector, visit order), each
// sector keeps its heaviest offset (first one among equals: the reference's stable bubble sort), the
// twelve winners are sorted by weight and the top eight stored.
// Launch shape (this function: more than 32 views, or another window than the default; else gen_candidates_views_px below):
// pixels x views (the view is wave-uniform), one pass over the window per lane, the twelve
// winners in registers.  (Round 1 walked the window once per pixel for all views and kept S x 12 running
// maxima in a dynamically indexed private array: 3.2 KB of scratch per lane, 330 GB of write-back per
// launch at 6208x4128.)  For the default window (weak_radius = 5, main.h:104; the reference never
// changes it) the sector lists are compile-time constants (dvp_sector5.inc): every load has a static
// offset and no branch in front of it, so the compiler batches them.
DVP_HD void gen_candidates_px(const Dev& d, int px, int py, int v) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const float* ref = d.images;
	const float cpix = img_texel(ref, d.org, d.pitch, W, H, px, py);
	struct Best { float w; int i, j; };
	struct alignas(16) Cand8 { s2 o[8]; };
	Best win[12];
	bool any = false;
#ifndef DVP_GEI_GENERIC
	if (P.weak_radius == 5) {
#else
	if (false) {
#endif
#pragma unroll
		for (int r = 0; r < 12; ++r) {
			Best b = Best{ 0.0f, 0, 0 };
			bool has = false;
#pragma unroll
			for (int t = 0; t < kSector5Max; ++t) {
				if (t >= kSector5Count[r]) continue;
				const int i = kSector5[r][t][0], j = kSector5[r][t][1];
				const int x = px + i, y = py + j;
				const bool in = x >= 0 && x < W && y >= 0 && y < H;
				// clamped address: the load is unconditional (in-bounds), its value is used only when `in`
				const uint32_t sv = d.selected_views[clampi(y, 0, H - 1) * W + clampi(x, 0, W - 1)];
				const float a = img_texel(ref, d.org, d.pitch, W, H, x, y);
				const float w = bilateral_weight((float)i, (float)j, a, cpix, P.sigma_spatial, P.sigma_color, 1);
				if (in && ((sv >> v) & 1) && (!has || w > b.w)) { has = true; b = Best{ w, i, j }; }
			}
			win[r] = b;
			any |= has;
		}
	} else {
		const int radius = P.weak_radius;
#pragma unroll
		for (int r = 0; r < 12; ++r) {
			Best b = Best{ 0.0f, 0, 0 };
			bool has = false;
			int cnt = 0;
			const int t1 = uniform_load_i32(d.sector_start, r + 1);
			for (int t = uniform_load_i32(d.sector_start, r); t < t1; ++t) {
				const int code = uniform_load_i32(d.sector_taps, t);
				const int i = (code & 0xffff) - radius, j = (code >> 16) - radius;
				const int x = px + i, y = py + j;
				if (!(x >= 0 && x < W && y >= 0 && y < H)) continue;
				if (!((d.selected_views[x + y * W] >> v) & 1)) continue;
				if (cnt >= 20) continue;   // regionCounts[region] < 20 (APD.cu:3768)
				cnt++;
				const float a = img_texel(ref, d.org, d.pitch, W, H, x, y);
				const float w = bilateral_weight((float)i, (float)j, a, cpix, P.sigma_spatial, P.sigma_color, 1);
				if (!has || w > b.w) { has = true; b = Best{ w, i, j }; }
			}
			win[r] = b;
			any |= has;
		}
	}
	if (any) {   // stable descending sort of the 12 sector winners (empty sectors: weight 0, offset (0,0))
#pragma unroll
		for (int a = 1; a < 12; ++a) {
#pragma unroll
			for (int b = a; b >= 1; --b) {
				const bool sw = win[b - 1].w < win[b].w;
				const Best lo = win[b - 1], hi = win[b];
				win[b - 1] = sw ? hi : lo;
				win[b] = sw ? lo : hi;
			}
		}
	}
	Cand8 rec;
#pragma unroll
	for (int k = 0; k < 8; ++k) rec.o[k] = mks2(win[k].i, win[k].j);
	*reinterpret_cast<Cand8*>(d.candidate + cand_index(d, center, v)) = rec;   // two 16-byte stores
}

// The same records for ALL source views of a pixel by one lane (round 6).  A tap's weight does not depend on the view — only
// whether the tap's pixel sees it does — so the pixels x views launch evaluated every weight (an exp among ~35 instructions) and
// fetched every texel and selected-view word S times (52 ms per 25-Mpx launch with S = 9).  Here: one pass over the window, the
// weight once, S running sector maxima in registers (views unrolled over MV, nothing indexed dynamically), and instead of twelve
// winners per view to sort at the end, each view's sorted top eight kept as the sectors finish: inserting the sectors' winners in
// sector order, each BELOW the entries that are not lighter, is the reference's stable bubble sort (APD.cu:3786-3792) step by step,
// and an entry that falls out of the first eight cannot come back.  Empty sectors take part with weight 0 and offset (0, 0) as they
// do there (a view no tap sees: eight zero offsets either way).  weak_radius == 5 (dvp_sector5.inc) only; a lane takes MV views, the
// launch has ceil(S / MV) lanes per pixel.
template <int MV>
DVP_HD void gen_candidates_views_px(const Dev& d, int px, int py, int v0 = 0) {   // views v0 .. v0 + MV - 1
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const int S = P.num_images - 1;
	const float* ref = d.images;
	const float cpix = img_texel(ref, d.org, d.pitch, W, H, px, py);
	float top_w[MV][8];
	uint32_t top_o[MV][8];   // offset as the record stores it: (uint16)i | (uint16)j << 16
#pragma unroll
	for (int v = 0; v < MV; ++v)
#pragma unroll
		for (int k = 0; k < 8; ++k) { top_w[v][k] = -1.0f; top_o[v][k] = 0u; }   // (-1: below every weight, an empty place)
#pragma unroll
	for (int r = 0; r < 12; ++r) {
		float bw[MV];
		uint32_t bo[MV];
		uint32_t has = 0;
#pragma unroll
		for (int v = 0; v < MV; ++v) { bw[v] = 0.0f; bo[v] = 0u; }
#pragma unroll
		for (int t = 0; t < kSector5Max; ++t) {
			if (t >= kSector5Count[r]) continue;
			const int i = kSector5[r][t][0], j = kSector5[r][t][1];
			const int x = px + i, y = py + j;
			const bool in = x >= 0 && x < W && y >= 0 && y < H;
			const uint32_t sv = d.selected_views[clampi(y, 0, H - 1) * W + clampi(x, 0, W - 1)];   // (in-bounds; used only when `in`)
			const float a = img_texel(ref, d.org, d.pitch, W, H, x, y);
			const float w = bilateral_weight((float)i, (float)j, a, cpix, P.sigma_spatial, P.sigma_color, 1);
			const uint32_t o = ((uint32_t)i & 0xffffu) | ((uint32_t)j << 16);
			const uint32_t seen = in ? (sv >> v0) : 0u;
#pragma unroll
			for (int v = 0; v < MV; ++v) {
				const bool take = ((seen >> v) & 1u) && (!((has >> v) & 1u) || w > bw[v]);
				bw[v] = take ? w : bw[v];
				bo[v] = take ? o : bo[v];
				has |= take ? (1u << v) : 0u;
			}
		}
		// the sector's winners into the views' lists
#pragma unroll
		for (int v = 0; v < MV; ++v) {
			const float w = bw[v];
			const uint32_t o = bo[v];
#pragma unroll
			for (int k = 7; k >= 1; --k) {
				const bool from_above = top_w[v][k - 1] < w;   // the entry above is lighter: it moves down to k
				const bool here = top_w[v][k] < w;              // (else) the new entry lands at k if k's is lighter
				top_o[v][k] = from_above ? top_o[v][k - 1] : (here ? o : top_o[v][k]);
				top_w[v][k] = from_above ? top_w[v][k - 1] : (here ? w : top_w[v][k]);
			}
			const bool first = top_w[v][0] < w;
			top_o[v][0] = first ? o : top_o[v][0];
			top_w[v][0] = first ? w : top_w[v][0];
		}
	}
#pragma unroll
	for (int v = 0; v < MV; ++v) {
		if (v0 + v >= S) continue;
		uint32_t* out = reinterpret_cast<uint32_t*>(d.candidate + cand_index(d, center, v0 + v));
#if defined(__HIP_DEVICE_COMPILE__)
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		u4 lo, hi;
		lo.x = top_o[v][0]; lo.y = top_o[v][1]; lo.z = top_o[v][2]; lo.w = top_o[v][3];
		hi.x = top_o[v][4]; hi.y = top_o[v][5]; hi.z = top_o[v][6]; hi.w = top_o[v][7];
		reinterpret_cast<u4*>(out)[0] = lo;
		reinterpret_cast<u4*>(out)[1] = hi;
#else
		for (int k = 0; k < 8; ++k) d.candidate[cand_index(d, center, v0 + v) + k] = mks2((int)(int16_t)(top_o[v][k] & 0xffffu), (int)(int16_t)(top_o[v][k] >> 16));
#endif
	}
}
#ifndef DVP_CAND_GROUP
#define DVP_CAND_GROUP 5   // 25-Mpx launch, S = 9: 42 ms (10: 110 ms at one wave per SIMD, 3: 47, 1: 97; a lane per (pixel, view) with a sort at the end: 53)
#endif
constexpr int kCandGroup = DVP_CAND_GROUP;   // views per lane (16 + 2 registers each); the launch has ceil(S / kCandGroup) lanes per pixel
constexpr int kCandViewsMax = 32;
DVP_HD bool gen_candidates_all_views(const Dev& d) {
#if defined(DVP_GEI_GENERIC) || defined(DVP_CAND_PER_VIEW)
	return false;
#else
	return d.params.weak_radius == 5 && d.params.num_images - 1 <= kCandViewsMax;
#endif
}

// the per-pixel rest of GenEdgeInform (APD.cu:3796-3890); candidates: gen_candidates_px, edge_neigh: edge_ray_line
DVP_HD void gen_edge_inform_px(const Dev& d, int px, int py) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	const int dxs[8] = { 0, 0, -1, 1, -1, 1, -1, 1 };
	const int dys[8] = { -1, 1, 0, 0, -1, 1, 1, -1 };
	if (P.use_edge) {
		// edge_neigh is filled by the line-scan pre-pass (edge_ray_line) of the same launch site
		if (d.weak_info[center] == DVP_WEAK) {
			const int radius = P.strong_radius;
			int edge_pix = 0, tot_pix = 0;
			for (int i = -radius; i <= radius; i++)
				for (int j = -radius; j <= radius; j++) {
					const int nx = px + i, ny = py + j;
					if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;
					if (d.edge[ny * W + nx]) edge_pix++;
					tot_pix++;
				}
			const float density = 1.0f * edge_pix / tot_pix;
			d.complex_[d.neighbours_map[center]] = 1.0f / (1.0f + dvp_expf(-25.0f * (density - 0.35f)));
		}
		if (P.state == DVP_REFINE_INIT && P.use_detail && d.edge[center]) {
			if (d.weak_info[center] != DVP_STRONG) d.weak_info[center] = DVP_UNKNOWN;
		}
	}
	if (P.use_label && d.weak_info[center] == DVP_WEAK) {
		s2* lb = d.label_boundary + (size_t)d.neighbours_map[center] * 8;
		const int cl = d.label[center];
		if (cl > 0) {
			for (int k = 0; k < 8; k++) {
				// The reference walks from the pixel to the first label -1 (or the border) and keeps the LAST
				// pixel of its own label it passed (APD.cu:3861-3873): thousands of dependent loads per ray
				// on a large region.  Same answer from the other end: the line-scan pre-pass (edge_ray_line,
				// what = 1) gives the stop pixel, and the farthest pixel of the own label is the first one
				// met walking BACK from there — on a region bounded by edges that is the first probe.
				const s2 stop = d.label_stop[(size_t)center * 8 + k];
				int nx, ny;   // last pixel before the stop (or the last pixel inside the image)
				if (stop.x != -1) { nx = stop.x - dxs[k]; ny = stop.y - dys[k]; }
				else {
					int steps = 1 << 30;   // pixels from (px,py) to the border along k
					if (dxs[k] > 0) steps = DVP_MIN(steps, W - 1 - px); else if (dxs[k] < 0) steps = DVP_MIN(steps, px);
					if (dys[k] > 0) steps = DVP_MIN(steps, H - 1 - py); else if (dys[k] < 0) steps = DVP_MIN(steps, py);
					nx = px + steps * dxs[k];
					ny = py + steps * dys[k];
				}
				int lx = -1, ly = -1;
				while (nx != px || ny != py) {
					if (d.label[nx + ny * W] == cl) { lx = nx; ly = ny; break; }
					nx -= dxs[k];
					ny -= dys[k];
				}
				lb[k] = mks2(lx, ly);
			}
		}
		if (P.state == DVP_REFINE_INIT && P.use_detail && d.label[center] == 0) {
			if (d.weak_info[center] != DVP_STRONG) d.weak_info[center] = DVP_UNKNOWN;
		}
	}
}

// ---- FindNearestStrongPoint (APD.cu:4159-4193) -------------------------------------------------
DVP_HD void find_nearest_strong_px(const Dev& d, int px, int py) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	s2 res = mks2(-1, -1);
	if (d.weak_info[center] == DVP_WEAK) {
		// The reference scans the whole (2r+1)^2 square of every ring and skips the interior (APD.cu:4176-4179):
		// O(r^3) dependent loads.  Its visiting order on a ring is x ascending — the whole left column (y
		// ascending), then per inner column the top pixel before the bottom one, then the whole right column — so
		// the first hit is: the first set bit of the left column segment, else the smaller of the first set bits
		// of the top and bottom row segments (top wins a tie), else the first set bit of the right column segment.
		// Row segments come from the row-major bit tiles of (weak_info == STRONG), column segments from the
		// transposed tiles: a ring costs 4-8 word loads instead of 8r byte loads.
		for (int radius = 1; radius <= 100; ++radius) {   // (ring 0 is the WEAK pixel itself)
			const int xl = px - radius, xr = px + radius, yt = py - radius, yb = py + radius;
			const int ya = DVP_MAX(yt, 0), yz = DVP_MIN(yb, H - 1);
			if (xl >= 0) {
				const int y = first_set_bit_in<true>(d.strong_bits_t, d.edge_tiles_x, xl, ya, yz);
				if (y >= 0) { res = mks2(xl, y); break; }
			}
			const int xa = DVP_MAX(xl + 1, 0), xz = DVP_MIN(xr - 1, W - 1);
			if (xa <= xz) {
				const int xt = yt >= 0 ? first_set_bit_in<false>(d.strong_bits, d.edge_tiles_x, yt, xa, xz) : -1;
				const int xb = yb < H ? first_set_bit_in<false>(d.strong_bits, d.edge_tiles_x, yb, xa, xz) : -1;
				if (xt >= 0 && (xb < 0 || xt <= xb)) { res = mks2(xt, yt); break; }
				if (xb >= 0) { res = mks2(xb, yb); break; }
			}
			if (xr < W) {
				const int y = first_set_bit_in<true>(d.strong_bits_t, d.edge_tiles_x, xr, ya, yz);
				if (y >= 0) { res = mks2(xr, y); break; }
			}
		}
	}
	d.weak_nearest_strong[center] = res;
}

// ---- geometry predicates (APD.cu:244-311) -------------------------------------------------------
DVP_HD bool point_in_triangle(s2 A, s2 B, s2 C, int px, int py) {
	const f2 AB = mk2((float)(B.x - A.x), (float)(B.y - A.y));
	const f2 BC = mk2((float)(C.x - B.x), (float)(C.y - B.y));
	const f2 CA = mk2((float)(A.x - C.x), (float)(A.y - C.y));
	const float ab = sqrtf(AB.x * AB.x + AB.y * AB.y);
	const float bc = sqrtf(BC.x * BC.x + BC.y * BC.y);
	const float ca = sqrtf(CA.x * CA.x + CA.y * CA.y);
	if (ab <= 2 || bc <= 2 || ca <= 2) return false;
	if (!(ab + bc > ca && bc + ca > ab && ab + ca > bc)) return false;
	const f2 PA = mk2((float)(A.x - px), (float)(A.y - py));
	const f2 PB = mk2((float)(B.x - px), (float)(B.y - py));
	const f2 PC = mk2((float)(C.x - px), (float)(C.y - py));
	const float t1 = PA.x * PB.y - PA.y * PB.x;
	const float t2 = PB.x * PC.y - PB.y * PC.x;
	const float t3 = PC.x * PA.y - PC.y * PA.x;
	return t1 * t2 >= 0 && t1 * t3 >= 0;
}

// true = the segment B->A crosses an edge pixel (APD.cu:267-311) — the definition: every step of the reference's walk is made.
// The reference walks the line one pixel per loop iteration and tests the edge map after every step
// (one dependent load per step, up to max(W,H)/30 of them).  The answer is "any visited pixel is
// an edge pixel", so the walk is done in blocks of 8 steps: eight positions from the integer
// line state, eight independent loads from the bit-tiled edge map (dvp_dev.hpp: edge_bit), one OR.
// Visited set, step limit and the one-pixel
// overshoot past the end point (the loop condition is tested after the step) are the reference's.
DVP_HD bool bresenham_hits_edge_steps(const Dev& d, int Ax, int Ay, int Bx, int By) {
	const int W = d.width, H = d.height;
	const int max_step = (int)(DVP_MAX(H, W) / 30.0);
	int x0 = Bx, y0 = By;
	const int x1 = Ax, y1 = Ay;
	const int ABx = Ax - Bx, ABy = Ay - By;
	if (ABx * ABx + ABy * ABy > 9 * max_step * max_step) return false;
	// Every pixel the walk can visit lies in the end points' bounding box widened by two pixels: the major
	// axis advances on every step and keeps going while the minor axis catches up, and the exit is tested
	// after the step (exhaustive check over all |dx|, |dy| < 260 and random ones up to 3 x 1092 steps: the walk
	// leaves the box by at most 2).  If the widened box holds no edge pixel at all — four loads from the cell
	// table — neither end point is an edge pixel and the walk cannot hit.
	if (edge_count_upper(d, DVP_MIN(x0, x1) - 2, DVP_MIN(y0, y1) - 2, DVP_MAX(x0, x1) + 2, DVP_MAX(y0, y1) + 2) == 0) return false;
	if (edge_bit(d, x0, y0) || edge_bit(d, x1, y1)) return false;
	const int dx = x1 > x0 ? x1 - x0 : x0 - x1, sx = x0 < x1 ? 1 : -1;
	const int dy = y1 > y0 ? y1 - y0 : y0 - y1, sy = y0 < y1 ? 1 : -1;
	int erro = (dx > dy ? dx : dy) / 2;
	int step = 0;
	bool tagx = true, tagy = true;
	bool alive = true;       // the reference's loop would still be running
	while (alive) {
		int wi[8], sh[8];   // word index in the bit-tiled map (-1: outside the image), bit
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			wi[k] = -1;
			sh[k] = 0;
			if (alive) {
				if (x0 == x1) tagx = false;
				if (y0 == y1) tagy = false;
				const int e2 = erro;
				if (e2 > -dx) { erro -= dy; x0 += sx; }
				if (e2 < dy) { erro += dx; y0 += sy; }
				if (x0 >= 0 && x0 < W && y0 >= 0 && y0 < H) {
					wi[k] = tile_word(d.edge_tiles_x, x0, y0);
					sh[k] = x0 & 31;
				}
				step += 1;
				if (step >= max_step || !(tagx || tagy)) alive = false;
			}
		}
		unsigned hit = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) hit |= (d.edge_bits[wi[k] < 0 ? 0 : wi[k]] >> sh[k]) & (wi[k] < 0 ? 0u : 1u);
		if (hit) return true;
	}
	return false;
}

// The same answer without making every step (round 5).  The reference's walk — e2 = err; if (e2 > -dx) { err -= dy; x += sx; }
// if (e2 < dy) { err += dx; y += sy; }, err0 = max(dx, dy) / 2 — is autonomous: its state after k iterations has a closed
// form (checked against the loop for all dx, dy < 120 and k <= 300, and through the engine's own tests):
//   dx >= dy:  x steps every iteration,  y has stepped floor((k dy + dx - 1 - err0) / dx) times,  err_k = dx - 1 - remainder
//   dx <  dy:  y steps every iteration,  x has stepped min(k, q), q = floor((err0 + k dx + dy - 1) / dy) times
// and it tests the positions after iterations 1 .. N, N = min(max(dx, dy) + 1, max_step) (both end-point flags are down at
// the start of iteration max(dx, dy), which still makes its step: the one-pixel overshoot).  Both coordinates are monotone
// along the walk, so the eight positions of a block lie in the box of its first and last: a block whose box (widened to
// whole 8 x 8 cells: the cell table, four loads) holds no edge pixel is skipped WITHOUT stepping through it; a block that may
// hit restarts the reference's stepping from the closed-form state.  GenNeighbours' searches spent 36 + 14 of their 113 ms per
// cfg3 pass in these walks (tools/gn_ablate.sh), most of it integer stepping through edge-free pixels on the way to the
// one edge that blocks a direction.
DVP_HD int line_div(int n, int dvs, float inv) {   // floor(n / dvs) for 0 <= n < 2^23, dvs >= 1: one multiplication + fix-up
	int q = (int)((float)n * inv);
	const int r = n - q * dvs;
	q += (r >= dvs) ? 1 : 0;
	q -= (r < 0) ? 1 : 0;
	return q;
}
DVP_HD bool bresenham_hits_edge(const Dev& d, int Ax, int Ay, int Bx, int By) {
#if defined(DVP_WALK_STEPS)   // A/B builds: the stepping definition
	return bresenham_hits_edge_steps(d, Ax, Ay, Bx, By);
#endif
	const int W = d.width, H = d.height;
	const int max_step = (int)(DVP_MAX(H, W) / 30.0);
	const int x0 = Bx, y0 = By;
	const int x1 = Ax, y1 = Ay;
	const int ABx = Ax - Bx, ABy = Ay - By;
	if (ABx * ABx + ABy * ABy > 9 * max_step * max_step) return false;
	if (edge_count_upper(d, DVP_MIN(x0, x1) - 2, DVP_MIN(y0, y1) - 2, DVP_MAX(x0, x1) + 2, DVP_MAX(y0, y1) + 2) == 0) return false;
	if (edge_bit(d, x0, y0) || edge_bit(d, x1, y1)) return false;
	const int dx = x1 > x0 ? x1 - x0 : x0 - x1, sx = x0 < x1 ? 1 : -1;
	const int dy = y1 > y0 ? y1 - y0 : y0 - y1, sy = y0 < y1 ? 1 : -1;
	if (dx == 0 && dy == 0) return false;   // the walk never leaves its first pixel, which is not an edge pixel
	const bool xmaj = dx >= dy;
	const int major = xmaj ? dx : dy;
	const int e0 = major / 2;
	const int N = DVP_MIN(major + 1, max_step);
	if (N <= 0) return false;
	const float inv = 1.0f / (float)major;
	// state after k iterations: steps along x and y, the error term
	auto state = [&](int k, int* a, int* b, int* e) {
		if (xmaj) {
			const int n = k * dy + dx - 1 - e0, q = line_div(n, dx, inv);
			*a = k; *b = q; *e = dx - 1 - (n - q * dx);
		} else {
			const int n = e0 + k * dx + dy - 1, q = line_div(n, dy, inv);
			*b = k;
			if (q < k) { *a = q; *e = (n - q * dy) - dy + 1; }
			else { *a = k; *e = e0 + k * (dx - dy); }
		}
	};
	for (int k0 = 1; k0 <= N; k0 += 8) {
		const int k1 = DVP_MIN(k0 + 7, N);
		int a0, b0, e, a1, b1, e1, ap, bp;
		state(k0 - 1, &ap, &bp, &e);   // where the block's first iteration starts
		state(k1, &a1, &b1, &e1);
		{   // first position of the block: one iteration from (ap, bp, e)
			a0 = ap + ((e > -dx) ? 1 : 0);
			b0 = bp + ((e < dy) ? 1 : 0);
		}
		const int xa = x0 + sx * a0, ya = y0 + sy * b0, xb = x0 + sx * a1, yb = y0 + sy * b1;
		if (edge_count_upper(d, DVP_MIN(xa, xb), DVP_MIN(ya, yb), DVP_MAX(xa, xb), DVP_MAX(ya, yb)) == 0) continue;
		// the reference's steps k0 .. k1 from the state after k0 - 1 iterations
		int x = x0 + sx * ap, y = y0 + sy * bp, erro = e;
		int wi[8], sh[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			wi[j] = -1;
			sh[j] = 0;
			if (k0 + j <= k1) {
				const int e2 = erro;
				if (e2 > -dx) { erro -= dy; x += sx; }
				if (e2 < dy) { erro += dx; y += sy; }
				if (x >= 0 && x < W && y >= 0 && y < H) {
					wi[j] = tile_word(d.edge_tiles_x, x, y);
					sh[j] = x & 31;
				}
			}
		}
		unsigned hit = 0;
#pragma unroll
		for (int j = 0; j < 8; ++j) hit |= (d.edge_bits[wi[j] < 0 ? 0 : wi[j]] >> sh[j]) & (wi[j] < 0 ? 0u : 1u);
		if (hit) return true;
	}
	return false;
}

// ---- GenNeighbours (APD.cu:3330-3711) -----------------------------------------------------------
// First third, one lane per WEAK pixel: the directional search (APD.cu:3382-3452).  The search of a direction depends on
// the points found for the directions before it and on how many random tries those consumed, so a pixel is sequential
// here, and the lanes of a wave — neighbouring WEAK pixels — walk the SAME direction at the same time: their tries land
// near each other (measured: letting every lane run through its directions at its own pace costs 50 ms of 80, a wave per
// pixel with the tries over the lanes 6-25 ms).  The list of found points lives in LDS (`pts`, one
// column per lane): as a private array it is scratch memory, re-read for every try's duplicate test — 100 GB of HBM
// fetch per cfg3 launch (FETCH_SIZE, profiles/pmc_r03.json) for 0.2 GB of state.  The label extension (:3455-3560)
// and the plane fit (:3562-3711) are one wave per pixel: gen_neighbours_extend_wave / gen_neighbours_fit_wave.
constexpr int kGnMaxPoints = 160;   // max_pt_num (APD.cu:3338)
constexpr int kGnDirSlots = 32;     // 8 octants x 4 rotations: the directional slots at the head of the list
DVP_HD void gen_neighbours_px(const Dev& d, int px, int py, s2* pts, int stride) {
	const int W = d.width, H = d.height;
	const int center = px + py * W;
	if (d.weak_info[center] != DVP_WEAK) return;
	const DvpParams& P = d.params;
	const int min_margin = 6;
	const int wi = d.neighbours_map[center];
	s2* neighbours = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	Rng r_limit(d.seed, (uint32_t)center, rng_site(PH_NEIGHBOURS, 0, SUB_LIMIT));
	Rng r_search(d.seed, (uint32_t)center, rng_site(PH_NEIGHBOURS, 0, SUB_SEARCH));

	for (int i = 0; i < DVP_NEIGHBOUR_NUM; ++i) neighbours[i] = mks2(-1, -1);
	neighbours[0] = mks2(px, py);
	// (-1,-1) == not valid (the reference's dir_valid[])
	for (int i = 0; i < kGnDirSlots; ++i) pts[i * stride] = mks2(-1, -1);
	int strong_point_size = 0;
	const int rotate_time = P.rotate_time;

	bool edge_limit = false;
	if (P.use_limit) {
		edge_limit = true;
		if (P.use_edge) {
			const float complex_val = d.complex_[wi];
			const float rp = r_limit.uniform() - FLT_EPSILON;
			if (rp < complex_val) edge_limit = false;
		}
	}

	int odi = -1;
	for (int odx = -1; odx <= 1; ++odx) {
		for (int ody = -1; ody <= 1; ++ody) {
			if (odx == 0 && ody == 0) continue;
			f2 od = mk2((float)odx, (float)ody);
			normalize2(&od);
			odi++;
			for (int rot = 0; rot < rotate_time; ++rot) {
				const int dir_index = odi * 4 + rot;
				// The (radius, try) double loop of APD.cu:3396-3452 as a state machine in two alternating
				// phases: (A) advance through the tries until one passes the cheap tests, (B) the line walk
				// of that try for all lanes of the wave together.  Try order and RNG consumption per lane
				// are unchanged.
				bool found_dir = false, out = false;
				int radius = 2, ri = 0;
				while (!found_dir && !out) {
					bool cand = false;
					s2 np = mks2(-1, -1);
					while (!out) {
						if (ri == 0) {   // entering a new radius: the ray must still be inside the image
							const float tx = px + od.x * radius, ty = py + od.y * radius;
							if (tx < 0 || ty < 0 || tx >= W || ty >= H) { out = true; break; }
						}
						const int cur_radius = radius;
						// advance the state first: the try below may leave the inner loop
						if (++ri == 4) {
							ri = 0;
							radius = DVP_MIN(radius * 2, radius + 25);
							if (radius > 4096) out = true;
						}
						const uint32_t sgx = (r_search.next() % 2 == 0) ? 1u : 0xFFFFFFFFu;
						const int xs = (int)((sgx * r_search.next()) % (uint32_t)d.nb_shift_range);
						const uint32_t sgy = (r_search.next() % 2 == 0) ? 1u : 0xFFFFFFFFu;
						const int ys = (int)((sgy * r_search.next()) % (uint32_t)d.nb_shift_range);
						f2 dir = mk2(od.x * 20 + xs, od.y * 20 + ys);
						normalize2(&dir);
						np = mks2((int)(px + dir.x * cur_radius), (int)(py + dir.y * cur_radius));
						if (np.x < min_margin || np.y < min_margin || np.x >= W - min_margin || np.y >= H - min_margin) continue;
						int npc = np.x + np.y * W;
						if (!strong_bit(d, np.x, np.y)) {   // weak_info[npc] != STRONG, from the L2-resident bit map
							np = d.weak_nearest_strong[npc];
							if (np.x == -1 || np.y == -1) continue;
							npc = np.x + np.y * W;
						}
						// the angle test before the duplicate test (the reference has them the other way round, APD.cu:3425-3440;
						// both only skip the try): the cheap one first
						f2 td = mk2((float)(np.x - px), (float)(np.y - py));
						normalize2(&td);
						const float cos_a = td.x * od.x + td.y * od.y;
						if (!(cos_a > d.nb_thresh)) continue;
						bool same = false;   // (no early exit: the loads are independent and pipeline)
						for (int k = 0; k < dir_index; k++) {
							const s2 q = pts[k * stride];
							same |= (q.x == np.x) & (q.y == np.y);
						}
						if (same) continue;
						cand = true;
						break;
					}
					if (!cand) break;
					if (!edge_limit || !bresenham_hits_edge(d, px, py, np.x, np.y)) {
						pts[dir_index * stride] = np;
						strong_point_size++;
						found_dir = true;
					}
				}
				f2 rd;
				rd.x = od.x * d.nb_cos - od.y * d.nb_sin;
				rd.y = od.x * d.nb_sin + od.y * d.nb_cos;
				normalize2(&rd);
				od = rd;
			}
		}
	}

	// hand-over to the wave-per-pixel rest (dvp_weak_wave.hpp): the 32 directional slots with their holes, and how many are filled
	s2* out = d.gn_points + (size_t)wi * kGnDirSlots;
	for (int i = 0; i < kGnDirSlots; ++i) out[i] = pts[i * stride];
	d.gn_count[wi] = strong_point_size;
}

// NeigbourUpdate (APD.cu:3713-3729)
DVP_HD void neighbour_update_px(const Dev& d, int px, int py) {
	const int center = px + py * d.width;
	if (d.weak_info[center] != DVP_WEAK) return;
	if (d.weak_reliable[center] != 1) d.weak_info[center] = DVP_UNKNOWN;
}

// ---- RANSACToGetFitPlane (APD.cu:4195-4404) -----------------------------------------------------
// the adaptive patch radius RANSACToGetFitPlane leaves (APD.cu:4352-4402): from the area of the winning triangle (Heron), capped by the
// nearest of its corners, by the nearest edge pixel of the eight rays and by the nearest label boundary; rounded down to 2r % 5 == 0
DVP_HD int ransac_patch_radius(const Dev& d, int px, int py, int center, s2 A, s2 B, s2 C, bool edge_limit) {
	const DvpParams& P = d.params;
	const float a = sqrtf((float)((A.x - B.x) * (A.x - B.x) + (A.y - B.y) * (A.y - B.y)));
	const float b = sqrtf((float)((B.x - C.x) * (B.x - C.x) + (B.y - C.y) * (B.y - C.y)));
	const float c = sqrtf((float)((C.x - A.x) * (C.x - A.x) + (C.y - A.y) * (C.y - A.y)));
	const float pp = (float)((a + b + c) / 2.0);
	const float Sa = sqrtf(pp * (pp - a) * (pp - b) * (pp - c));
	const double rr = floor(sqrtf(Sa) / 2.0);
	int radius = (rr == rr) ? (int)rr : 0;
	const float Ad = sqrtf((float)((A.x - px) * (A.x - px) + (A.y - py) * (A.y - py)));
	const float Bd = sqrtf((float)((B.x - px) * (B.x - px) + (B.y - py) * (B.y - py)));
	const float Cd = sqrtf((float)((C.x - px) * (C.x - px) + (C.y - py) * (C.y - py)));
	const float min_dis = DVP_MIN(DVP_MIN(Ad, Bd), Cd);
	if (2.5 * min_dis < radius) radius = (int)min_dis;
	if (edge_limit) {
		if (P.use_edge) {
			float med = FLT_MAX;
			const s2* en = d.edge_neigh + (size_t)center * 8;
			for (int k = 0; k < 8; ++k) {
				const s2 ep = en[k];
				if (ep.x == -1 || ep.y == -1) continue;
				const float dist = sqrtf((float)((ep.x - px) * (ep.x - px) + (ep.y - py) * (ep.y - py)));
				med = DVP_MIN(med, dist);
			}
			if (med < radius) radius = (int)med;
		}
		if (P.use_label) {
			float mbd = FLT_MAX;
			const s2* lb = d.label_boundary + (size_t)d.neighbours_map[center] * 8;
			for (int k = 0; k < 8; ++k) {
				const s2 bp = lb[k];
				if (bp.x == -1 || bp.y == -1) continue;
				const double ex = (double)(px - bp.x), ey = (double)(py - bp.y);
				const float dist = (float)sqrt(ex * ex + ey * ey);
				mbd = DVP_MIN(mbd, dist);
			}
			if (mbd < radius) radius = (int)mbd;
		}
	}
	if (radius < 0) radius = 0;
	while ((radius << 1) % 5 != 0) radius--;
	return radius;
}

DVP_HD void ransac_fit_plane_px(const Dev& d, int px, int py, int iter) {
	const int W = d.width;
	const int center = px + py * W;
	const DvpParams& P = d.params;
	if (d.weak_info[center] != DVP_WEAK) { d.fit_planes[center] = d.planes[center]; return; }
	const DvpCamera cam = load_camera(d, 0);
	Rng r_limit(d.seed, (uint32_t)center, rng_site(PH_RANSAC, iter, SUB_LIMIT));
	Rng r_ransac(d.seed, (uint32_t)center, rng_site(PH_RANSAC, iter, SUB_RANSAC));
	bool edge_limit = false;
	if (P.use_limit) {
		edge_limit = true;
		if (P.use_edge) {
			const float complex_val = d.complex_[d.neighbours_map[center]];
			const float rp = r_limit.uniform() - FLT_EPSILON;
			if (rp < complex_val) edge_limit = false;
		}
	}
	s2 sp[11];
	f3 sp3[11], spn[11];
	int cnt = 0;
	const s2* nbs = d.neighbours + (size_t)d.neighbours_map[center] * DVP_NEIGHBOUR_NUM;
	for (int i = 1; i < DVP_NEIGHBOUR_NUM; ++i) {
		const s2 tp = nbs[i];
		if (tp.x == -1 || tp.y == -1) continue;
		sp[cnt] = tp;
		const f4 pl = d.planes[tp.x + tp.y * W];
		const float depth = depth_from_plane(cam, pl, tp.x, tp.y);
		float X[3];
		get_3d_point(cam, tp.x, tp.y, depth, X);
		sp3[cnt] = mk3(X[0], X[1], X[2]);
		spn[cnt] = mk3(pl.x, pl.y, pl.z);
		cnt++;
	}
	if (cnt < 3) { d.fit_planes[center] = d.planes[center]; return; }

	int use_a = 0, use_b = 0, use_c = 0;
	float min_cost = FLT_MAX;
	f4 best_plane = mk4(0, 0, 0, 0);
	bool has_best = false;
	// The 50 draws (APD.cu:4262-4330) in two alternating phases so that the lanes of a wave do the
	// expensive part together: (A) each lane advances through its own draws until one survives the
	// cheap rejections (distinct indices, normals, pixel inside the triangle); (B) every lane that has
	// a candidate runs the line walks, the plane fit and the residual sum.  Same draws, same order,
	// same tests per lane — only the interleaving across lanes changes (a draw that fails a cheap
	// test no longer makes 63 other lanes wait through the walks of the one lane that passed).
	int it = 0;
	uint64_t memo_lo = 0, memo_hi = 0;
	for (;;) {
		bool cand = false;
		int ai = 0, bi = 0, ci = 0;
		while (it < 50) {
			++it;
			ai = (int)(r_ransac.next() % (uint32_t)cnt);
			bi = (int)(r_ransac.next() % (uint32_t)cnt);
			ci = (int)(r_ransac.next() % (uint32_t)cnt);
			if (ai == bi || bi == ci || ai == ci) continue;
			const f3 AN = spn[ai], BN = spn[bi], CN = spn[ci];
			if (AN.x * BN.x + AN.y * BN.y + AN.z * BN.z < 0.9f || AN.x * CN.x + AN.y * CN.y + AN.z * CN.z < 0.9f ||
				BN.x * CN.x + BN.y * CN.y + BN.z * CN.z < 0.9f) continue;
			if (!point_in_triangle(sp[ai], sp[bi], sp[ci], px, py)) continue;
			cand = true;
			break;
		}
		if (!cand) break;
		if (edge_limit) {
			// edge_test[11][11] of the reference (APD.cu:4260, 4283-4299): symmetric, first evaluation wins;
			// 55 unordered pairs x 2 bits in two registers
			const int pa[3] = { ai, bi, ci }, pb[3] = { bi, ci, ai };
			bool hit = false;
#pragma unroll
			for (int e = 0; e < 3; ++e) {
				const int hi = pa[e] > pb[e] ? pa[e] : pb[e], lo = pa[e] > pb[e] ? pb[e] : pa[e];
				const int idx = hi * (hi - 1) / 2 + lo;   // < 55
				const int sh = (idx & 31) * 2;
				uint32_t st = (uint32_t)(((idx < 32) ? memo_lo : memo_hi) >> sh) & 3u;
				if (st == 0u) {
					st = bresenham_hits_edge(d, sp[pa[e]].x, sp[pa[e]].y, sp[pb[e]].x, sp[pb[e]].y) ? 1u : 2u;
					if (idx < 32) memo_lo |= (uint64_t)st << sh; else memo_hi |= (uint64_t)st << sh;
				}
				hit |= st == 1u;
			}
			if (hit) continue;
		}
		const f3 A = sp3[ai], B = sp3[bi], C = sp3[ci];
		const f3 AC = mk3(A.x - C.x, A.y - C.y, A.z - C.z);
		const f3 BC = mk3(B.x - C.x, B.y - C.y, B.z - C.z);
		f4 cv;
		cv.x = AC.y * BC.z - BC.y * AC.z;
		cv.y = -(AC.x * BC.z - BC.x * AC.z);
		cv.z = AC.x * BC.y - BC.x * AC.y;
		cv.w = 0.0f;
		if ((cv.x == 0 && cv.y == 0 && cv.z == 0) || cv.x != cv.x || cv.y != cv.y || cv.z != cv.z) continue;
		normalize3(&cv);
		cv.w = -(cv.x * A.x + cv.y * A.y + cv.z * A.z);
		float temp_cost = 0.0f;
		for (int si = 0; si < cnt; ++si) {
			if (si == ai || si == bi || si == ci) continue;
			const float fx = (sp[si].x - cam.K[2]) / cam.K[0];
			const float fy = (sp[si].y - cam.K[5]) / cam.K[4];
			const float fit_depth = -cv.w / (cv.x * fx + cv.y * fy + cv.z);
			temp_cost += fabsf(fit_depth - sp3[si].z);
		}
		if (temp_cost < min_cost) {
			min_cost = temp_cost;
			best_plane = cv;
			has_best = true;
			use_a = ai; use_b = bi; use_c = ci;   // the reference reads index -1 here (APD.cu:4349); see DESIGN.md quirks
		}
	}
	if (!has_best) {
		d.fit_planes[center] = mk4(0, 0, 0, 0);
		if (P.use_radius) d.radius[center] = P.strong_radius;
		return;
	}
	const float depth = depth_from_plane(cam, d.planes[center], px, py);
	const f4 vdir = view_direction(cam, px, py, depth);
	const float dp = best_plane.x * vdir.x + best_plane.y * vdir.y + best_plane.z * vdir.z;
	if (dp > 0) { best_plane.x = -best_plane.x; best_plane.y = -best_plane.y; best_plane.z = -best_plane.z; best_plane.w = -best_plane.w; }
	d.fit_planes[center] = best_plane;
	if (P.use_radius) {
		const int radius = ransac_patch_radius(d, px, py, center, sp[use_a], sp[use_b], sp[use_c], edge_limit);
		d.radius[center] = radius < P.strong_radius ? 0 : radius;
	}
}

}  // namespace dvp
#endif
