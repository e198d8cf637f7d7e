// dvp_weak_phased.hpp — the weak-pixel update (CheckerboardPropagationWeak + PlaneHypothesisRefinementWeak,
// APD.cu:2739-3089, 1897-2008) as EIGHT launches: evaluation launches with one wave per group of 1-4 WEAK pixels and decision
// launches with one LANE per WEAK pixel.
//
// Why.  weak_update_wave (dvp_weak_wave.hpp) keeps a pixel's whole update in one wave.  Its deformable-NCC items fill the
// lanes, but everything between them — the joint view selection, the geometric-consistency table, the adoption rules,
// the refinement hypotheses with their random normal, the bookkeeping of three phases — is per-pixel code that all 64
// lanes execute with identical values: 1/64 of the machine per instruction, and a chain of dependent round trips that
// four waves per SIMD cannot hide.  PMC (r04, cfg3): 29.8 k wave-level VALU instructions per WEAK pixel and launch, 38 %
// of the launch time left when both patches are stubbed out, 48 spilled VGPRs, lane utilisation 0.67.  The per-pixel code
// is scalar work: here it runs one pixel per lane, 64 pixels per instruction, in its own launches at full occupancy; the
// wave launches keep only what is parallel inside a pixel.
//
//   E0  wave   the anchors' planes (<= 8) against every source view                          -> ev
//   D1  lane   joint view selection, candidate costs (geometric term), best candidate        -> rec
//   E1  wave   the current plane and the fit plane against the selected views                -> ev
//   D2  lane   cost of the current plane, adoption of candidate / fit plane, the five refinement hypotheses
//   E2a wave   hypotheses in range against the FIRST selected view; which can still be adopted    -> ev
//   E2b wave   the survivors against the other selected views                                  -> ev
//   D3  lane   adoption of the hypotheses, the final plane
//   E3  lane   cost of the final plane with the plain bilateral NCC (the per-lane evaluator of the strong path)
//
// Between launches a pixel's state travels in a 256-byte record (WeakRec), its centre-patch table (36 x (w, w ref)) and
// its cost vectors (S x 8 floats, a view's eight candidates side by side: two 16-byte loads) — per WEAK pixel, indexed like Dev::neighbours.  Every floating-point operation is the
// one weak_update_wave performs, on the same operands in the same order: same bits (tests: the phased form, the one-wave
// form and the oracle agree launch site by launch site).  A WEAK pixel reads other pixels' state only from its anchors,
// which are STRONG (GenNeighbours) and which no weak update writes, so the launches of one update need no ordering between
// pixels.
#ifndef DVP_WEAK_PHASED_HPP_
#define DVP_WEAK_PHASED_HPP_

#include "dvp_weak_wave.hpp"
#include "dvp_strong.hpp"

namespace dvp {

struct alignas(16) WeakRec {
	f4 pl[8];              // planes the next evaluation launch takes (anchors' planes / current + fit plane / hypotheses)
	f4 plane_now;
	f4 cand;               // the plane of the best propagation candidate (anchor min_cost_idx)
	float cost_now, costs_center, depth_now, weight_norm;
	uint32_t pmask, vmask, sel_mask, sel_now;
	uint32_t flag;         // anchors 0..7 that exist and are STRONG
	int min_cost_idx;
	float fcost_min;
	uint32_t skip_refine;
	// centre-patch context of this update (colour-only weights at the pixel's own radius); the table is Dev::weak_ctab
	float sum_ref, sum_ref_ref, wsum;
	int radius, inc, fast;
	uint32_t pad[6];
};
static_assert(sizeof(WeakRec) == 256, "hand-over record of the phased weak update");

// lanes `pred` of the wave as a bit mask (bit = lane; only lanes < 32 may set pred).  Host emulation: the caller ORs the
// lanes' bits one after the other.
DVP_HD uint32_t wave_lane_bit(bool pred, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)__ballot(pred ? 1 : 0);
#else
	return pred ? 1u << lane : 0u;
#endif
}

// cost vectors of a WEAK pixel: [view][plane slot 0..7] — the eight candidates of a view are 32 contiguous bytes (the lane-per-pixel
// decision launches read them with two 16-byte loads: as 4-byte loads at a stride of S floats they were most of those launches'
// L2 requests)
DVP_HD float* weak_ev_of(const Dev& d, int wi) { return d.weak_ev + (size_t)wi * 8 * (size_t)(d.params.num_images - 1); }
DVP_HD int weak_ev_index(int q, int v) { return v * 8 + q; }
// the eight plane slots of view v
DVP_HD void weak_ev_load8(const float* ev, int v, float* out) {
#if defined(__HIP_DEVICE_COMPILE__)
	typedef float f4v __attribute__((ext_vector_type(4)));
	const f4v a = reinterpret_cast<const f4v*>(ev + v * 8)[0], b = reinterpret_cast<const f4v*>(ev + v * 8)[1];
	out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
#else
	for (int k = 0; k < 8; ++k) out[k] = ev[v * 8 + k];
#endif
}

// ---- the evaluation launches: one wave per GROUP of WEAK pixels ------------------------------------------------------------
// ComputeBilateralNCCNew (APD.cu:835-1021) for every (plane, view) PAIR the pixels of the group have to evaluate.  One pixel
// per wave (wave_ncc_new_tab) fills its rounds only in the propagation launch (<= 8 planes x S views = 72 pairs); the later
// ones hold 2 x ~5, 5 x 1 and (survivors) x ~4 pairs per pixel — a round of 64 lanes a third full behind a chain of dependent
// loads (record -> anchors -> their views -> table -> gathers) that a single pixel cannot hide: measured 0.23 ms per pair and
// 880 k pixels in the first launch against 0.7 and 1.7 in the other two.  Here the pairs of up to kGrp consecutive pixels of
// the WEAK list are numbered flat and taken kWeakPairs per batch; the work items of a batch — (pair, anchor) sub-patches,
// (pair, patch row) centre rows, pair totals — are numbered pair-fastest, so the lanes of a load instruction are the planes
// of one (pixel, view) first (they share the anchor's table record and neighbouring texels), then the next view / pixel.
// The arithmetic of an item is anchor_cost_tab / patch_row_sums / the totalling section of wave_ncc_new_tab, unchanged.
//   MODE 0  E0: the anchors' planes against every view; also builds the pixel's centre-patch context and leaves it
//   MODE 1  E1: the record's planes against its views
//   MODE 2  E2a: the hypotheses in range against the FIRST selected view; decides which survive
//   MODE 3  E2b: the survivors against the other selected views
constexpr int kGrp = 4;   // pixels per wave at most (E0: kGrpWide)
constexpr int kGrpWide = 2;
template <int GRP>
struct WeakGroupSharedT {
	// per pixel of the group
	int center[GRP];                        // pixel index, < 0: no pixel in this slot (filled by the caller)
	f2 ctab[GRP][kTaps * kTaps];            // centre patch: (w, w * ref) per tap, row-major
	s2 nbs[GRP][DVP_NEIGHBOUR_NUM];
	uint32_t asel[GRP][kAnchors];           // selected_views word of anchor k
	float sum_ref[GRP], sum_ref_ref[GRP], wsum[GRP];
	int px[GRP], py[GRP], wi[GRP], radius[GRP], inc[GRP], fast[GRP];
	uint32_t pmask[GRP], vmask[GRP], alive[GRP];
	// per pair of the batch
	float Hq[kWeakPairs][9];
	float rows[kWeakPairs][kTaps][3];        // centre-patch row sums (s_s, s_ss, s_rs); before the first batch of MODE 0: w * ref * ref per tap
	float acost[kWeakPairs][kAnchors];       // anchor cost, < 0: does not count
	uint32_t pair[kWeakPairs];               // pixel slot | view << 8 | plane << 16 | (centre projects inside the view) << 24
	// the (anchor, pair) items of the batch that really are sub-patches (the anchor exists and selected the view), in item order
	uint16_t live[kWeakPairs * kAnchors];
	int n_live;
};
struct CtabView { const f2* ctab; };

template <int SMP, int FMT, int MODE, int GRP>
DVP_HD void weak_group_eval(const Dev& d, int G, unsigned long long* nevals, WeakGroupSharedT<GRP>& sh) {
	const int W = d.width;
	const DvpParams& P = d.params;
	const int S = P.num_images - 1;
	const uint32_t all_views = (S >= 32) ? 0xFFFFFFFFu : ((1u << S) - 1u);
	// ---- the pixels of the group ------------------------------------------------------------------------------------------
	DVP_LANES(g) {
		if (g < GRP) {
			uint32_t pm = 0, vm = 0;
			int px = 0, py = 0, wi = 0;
			const int center = g < G ? sh.center[g] : -1;
			if (g >= G) sh.center[g] = -1;
			if (center >= 0) {
				py = center / W;
				px = center - py * W;
				wi = d.neighbours_map[center];
				if (MODE == 0) vm = all_views;   // (pm: the anchors that count, below)
				else {
					const WeakRec& rec = d.weak_rec[wi];
					pm = rec.pmask;
					vm = rec.vmask;
					if (MODE == 2) { if (pm && vm) vm = vm & (0u - vm); else pm = vm = 0u; }   // the first selected view
					if (MODE == 3) vm = vm & (vm - 1u);                                        // the others
					sh.radius[g] = rec.radius; sh.inc[g] = rec.inc; sh.fast[g] = rec.fast;
					sh.sum_ref[g] = rec.sum_ref; sh.sum_ref_ref[g] = rec.sum_ref_ref; sh.wsum[g] = rec.wsum;
				}
			}
			sh.px[g] = px; sh.py[g] = py; sh.wi[g] = wi;
			sh.pmask[g] = pm; sh.vmask[g] = vm; sh.alive[g] = 0u;
		}
	}
	wave_sync();
	DVP_LANES(l) {
		for (int it = l; it < G * DVP_NEIGHBOUR_NUM; it += 64) {
			const int g = it / DVP_NEIGHBOUR_NUM, i = it - g * DVP_NEIGHBOUR_NUM;
			sh.nbs[g][i] = sh.center[g] >= 0 ? d.neighbours[(size_t)sh.wi[g] * DVP_NEIGHBOUR_NUM + i] : mks2(-1, -1);
		}
	}
	wave_sync();
	DVP_LANES(l) {
		for (int it = l; it < G * kAnchors; it += 64) {
			const int g = it / kAnchors, k = it - g * kAnchors;
			const s2 nb = sh.nbs[g][k + 1];
			uint32_t sv = 0;
			if (!(nb.x == -1 || nb.y == -1)) {
				const int nbc = nb.x + nb.y * W;
				sv = d.selected_views[nbc];
				if (MODE == 0 && k < 8 && d.weak_info[nbc] == DVP_STRONG) {   // the anchors' planes (APD.cu:2771-2779)
					d.weak_rec[sh.wi[g]].pl[k] = d.planes[nbc];
					wave_bits_or(&sh.pmask[g], 1u << k);
				}
			}
			sh.asel[g][k] = sv;
		}
	}
	// ---- the centre-patch context: built (MODE 0: wave_patch_ctx per pixel, colour-only weights) or fetched --------------------
	if (MODE == 0) {
		float* caa = &sh.rows[0][0][0];
		DVP_LANES(l) {
			for (int it = l; it < G * kTaps * kTaps; it += 64) {
				const int g = it / (kTaps * kTaps), t = it - g * (kTaps * kTaps);
				if (sh.center[g] < 0) continue;
				int radius, inc;
				patch_geometry(d, sh.center[g], &radius, &inc);
				if (!(inc > 0 && (2 * radius) / inc + 1 == kTaps)) continue;
				const int px = sh.px[g], py = sh.py[g];
				const float cpix = ref_texel_t<FMT>(d, px, py);
				const int ty = t / kTaps, tx = t - ty * kTaps;
				const int i = -radius + tx * inc, j = -radius + ty * inc;
				const float a = ref_texel_t<FMT>(d, px + i, py + j);
				const float w = bilateral_weight((float)i, (float)j, a, cpix, P.sigma_spatial, P.sigma_color, 1);
				const float wa = w * a;
				sh.ctab[g][t] = mk2(w, wa);
				caa[it] = wa * a;
			}
		}
		wave_sync();
		DVP_LANES(g) {
			if (g < G && sh.center[g] >= 0) {
				int radius, inc;
				patch_geometry(d, sh.center[g], &radius, &inc);
				const int fast = (inc > 0 && (2 * radius) / inc + 1 == kTaps) ? 1 : 0;
				float sr = 0.0f, srr = 0.0f, ws = 0.0f;
				if (fast) {
					for (int ty = 0; ty < kTaps; ++ty) {
						float sr_row = 0.0f, srr_row = 0.0f, ws_row = 0.0f;
						for (int tx = 0; tx < kTaps; ++tx) {
							const f2 t = sh.ctab[g][ty * kTaps + tx];
							sr_row += t.y;
							srr_row += caa[g * (kTaps * kTaps) + ty * kTaps + tx];
							ws_row += t.x;
						}
						sr += sr_row;
						srr += srr_row;
						ws += ws_row;
					}
				}
				sh.radius[g] = radius; sh.inc[g] = inc; sh.fast[g] = fast;
				sh.sum_ref[g] = sr; sh.sum_ref_ref[g] = srr; sh.wsum[g] = ws;
				WeakRec& rec = d.weak_rec[sh.wi[g]];
				rec.flag = sh.pmask[g];
				rec.radius = radius; rec.inc = inc; rec.fast = fast;
				rec.sum_ref = sr; rec.sum_ref_ref = srr; rec.wsum = ws;
			}
		}
		DVP_LANES(l) {
			for (int it = l; it < G * kTaps * kTaps; it += 64) {
				const int g = it / (kTaps * kTaps), t = it - g * (kTaps * kTaps);
				if (sh.center[g] >= 0) d.weak_ctab[(size_t)sh.wi[g] * (kTaps * kTaps) + t] = sh.ctab[g][t];
			}
		}
	} else {
		DVP_LANES(l) {
			for (int it = l; it < G * kTaps * kTaps; it += 64) {
				const int g = it / (kTaps * kTaps), t = it - g * (kTaps * kTaps);
				if (sh.center[g] >= 0) sh.ctab[g][t] = d.weak_ctab[(size_t)sh.wi[g] * (kTaps * kTaps) + t];
			}
		}
	}
	wave_sync();
	// ---- the pairs, numbered flat over the group: pixel, then view, then plane ------------------------------------------------
	int base[GRP];
	int T = 0;
#pragma unroll
	for (int g = 0; g < GRP; ++g) {
		base[g] = T;
		T += __builtin_popcount(sh.pmask[g]) * __builtin_popcount(sh.vmask[g]);
	}
	const DvpCamera rc = load_camera(d, 0);   // (MODE 2)
	for (int b0 = 0; b0 < T; b0 += kWeakPairs) {
		const int nb = DVP_MIN(kWeakPairs, T - b0);
		const float inv_nb = 1.0f / (float)nb;
		// section 0: lane = pair: which (pixel, view, plane) it is, its homography
		DVP_LANES(l) {
			if (l < nb) {
				const int p = b0 + l;
				int g = 0, bg = 0;
#pragma unroll
				for (int i = 1; i < GRP; ++i)
					if (p >= base[i]) { g = i; bg = base[i]; }
				const uint32_t pm = sh.pmask[g], vm = sh.vmask[g];
				const int np = __builtin_popcount(pm);
				const int local = p - bg;
				const int vslot = small_div(local, 1.0f / (float)np), qi = local - vslot * np;
				const int v = nth_set_bit(vm, vslot), q = nth_set_bit(pm, qi);   // 0-based view, plane slot
				f4 pl;
				if (MODE == 0) { const s2 a = sh.nbs[g][q + 1]; pl = d.planes[a.x + a.y * W]; }
				else pl = d.weak_rec[sh.wi[g]].pl[q];
				const ViewConst vc = d.views[v + 1];
				float H[9];
				homography(vc, pl, H);
				const f2 pt = apply_homography(H, sh.px[g], sh.py[g]);
				const uint32_t inside = !(pt.x >= vc.fw || pt.x < 0.0f || pt.y >= vc.fh || pt.y < 0.0f) ? 1u : 0u;
#pragma unroll
				for (int i = 0; i < 9; ++i) sh.Hq[l][i] = H[i];
				sh.pair[l] = (uint32_t)g | ((uint32_t)v << 8) | ((uint32_t)q << 16) | (inside << 24);
			}
		}
		if (DVP_LANE0) sh.n_live = 0;
		wave_sync();
		// section 0b: the (anchor k, pair) items, pair fastest.  An anchor that does not exist does not count (-1); one that did
		// not select the view counts 2 where it projects inside (the reference's 0/0 path, anchor_cost_tab); the others — 71 % at
		// cfg3 — are sub-patches: listed in item order, so that the rounds below are full (lane utilisation of the propagation
		// launch 0.70 before) and the planes of one (pixel, view, anchor) still sit in adjacent lanes
		const int n_anchor = nb * kAnchors;
		for (int it0 = 0; it0 < n_anchor; it0 += 64) {
			DVP_LANES(l) {
				const int it = it0 + l;
				bool live = false;
				if (it < n_anchor) {
					const int k = small_div(it, inv_nb), pi = it - k * nb;
					const uint32_t desc = sh.pair[pi];
					if (desc >> 24) {
						const int g = (int)(desc & 255u), v = (int)((desc >> 8) & 255u);
						const s2 nbk = sh.nbs[g][k + 1];
						if (nbk.x == -1 || nbk.y == -1) sh.acost[pi][k] = -1.0f;
						else if (is_set(sh.asel[g][k], v)) live = true;
						else {
							float H[9];
#pragma unroll
							for (int i = 0; i < 9; ++i) H[i] = sh.Hq[pi][i];
							const f2 nsp = apply_homography(H, nbk.x, nbk.y);
							const bool outside = nsp.x < 0 || nsp.y < 0 || nsp.x >= W || nsp.y >= d.height;
							sh.acost[pi][k] = outside ? -1.0f : 2.0f;
						}
					}
				}
				const int slot = wave_ordered_slot(live, &sh.n_live);
				if (live) sh.live[slot] = (uint16_t)it;
			}
		}
		wave_sync();
		// section 1: the sub-patches; then centre items (patch row, pair)
		DVP_LANES(l) {
			const int n_live = sh.n_live;
			for (int it0 = 0; it0 < n_live; it0 += 64) {
				if (it0 + l >= n_live) continue;
				const int it = sh.live[it0 + l];
				const int k = small_div(it, inv_nb), pi = it - k * nb;
				const uint32_t desc = sh.pair[pi];
				const int g = (int)(desc & 255u), v = (int)((desc >> 8) & 255u);
				float H[9];
#pragma unroll
				for (int i = 0; i < 9; ++i) H[i] = sh.Hq[pi][i];
				sh.acost[pi][k] = anchor_cost_tab<SMP, FMT>(d, H, img_plane<FMT>(d, v + 1), sh.nbs[g][k + 1], 2, d.anchor_tab + anchor_rec_index(d, sh.wi[g], v, k));
			}
			const int n_centre = nb * kTaps;
			for (int it0 = 0; it0 < n_centre; it0 += 64) {
				const int it = it0 + l;
				if (it >= n_centre) continue;
				const int r = small_div(it, inv_nb), pi = it - r * nb;
				const uint32_t desc = sh.pair[pi];
				if (!(desc >> 24)) continue;
				const int g = (int)(desc & 255u), v = (int)((desc >> 8) & 255u);
				const int fast = sh.fast[g];
				if (!fast && r != 0) continue;
				float Hc[9];
#pragma unroll
				for (int i = 0; i < 9; ++i) Hc[i] = sh.Hq[pi][i];
				if (fast) {
					float o[3];
					const CtabView cv{ sh.ctab[g] };
					patch_row_sums<SMP, FMT>(d, cv, Hc, img_plane<FMT>(d, v + 1), sh.px[g], sh.py[g], sh.radius[g], sh.inc[g], r, o);
					sh.rows[pi][r][0] = o[0];
					sh.rows[pi][r][1] = o[1];
					sh.rows[pi][r][2] = o[2];
				} else {   // generic (non-6-tap) patches sample the float planes
					sh.rows[pi][0][0] = ncc_patch_generic(d, Hc, d.images + (size_t)(v + 1) * d.plane_stride * 2, sh.px[g], sh.py[g], sh.radius[g], sh.inc[g], 1);
				}
			}
		}
		wave_sync();
		// section 2: lane = pair: rows and anchors summed in the reference's order -> the pixel's cost vectors
		DVP_LANES(l) {
			if (l < nb) {
				const uint32_t desc = sh.pair[l];
				const int g = (int)(desc & 255u), v = (int)((desc >> 8) & 255u), q = (int)((desc >> 16) & 255u);
				float out = 2.0f;
				if (desc >> 24) {
					float cc;
					if (sh.fast[g]) {
						float s_s = 0.0f, s_ss = 0.0f, s_rs = 0.0f;
						for (int r = 0; r < kTaps; ++r) {
							s_s += sh.rows[l][r][0];
							s_ss += sh.rows[l][r][1];
							s_rs += sh.rows[l][r][2];
						}
						cc = ncc_from_sums(sh.sum_ref[g], sh.sum_ref_ref[g], s_s, s_ss, s_rs, sh.wsum[g]);
					} else {
						cc = sh.rows[l][0][0];
					}
					float scost = 0.0f, scnt = 0.0f;
					for (int k = 0; k < kAnchors; ++k) {
						const float ac = sh.acost[l][k];
						if (ac >= 0.0f) { scost += ac; scnt += 1.0f; }
					}
					out = cc;
					if (scnt > 0.0f) {
						float sc2 = scost / scnt;   // strong_cost /= strong_count (int -> float, exact)
						sc2 = DVP_MIN(sc2, 2.0f);
						out = (float)(0.25 * cc + 0.75 * sc2);
					}
				}
				weak_ev_of(d, sh.wi[g])[weak_ev_index(q, v)] = out;
				if (MODE == 2) {
					// The weighted sum only grows (weights > 0, costs >= 0, IEEE addition and division are monotone), so a hypothesis
					// whose FIRST selected view alone is not below the best cost at entry can never be adopted (APD.cu:1361-1383).
					const WeakRec& rec = d.weak_rec[sh.wi[g]];
					const int w = d.view_weight[(size_t)sh.center[g] * 32 + v];
					float tc = 0.0f;
					if (P.geom_consistency) tc += w * (out + P.geom_factor * geom_cost_cams(d, rc, d.cameras[v + 1], v + 1, sh.px[g], sh.py[g], rec.pl[q]));
					else tc += w * out;
					if (tc / rec.weight_norm < rec.cost_now) wave_bits_or(&sh.alive[g], 1u << q);
				}
			}
		}
		wave_sync();
	}
	if (MODE == 2) {
		DVP_LANES(g) { if (g < G && sh.pmask[g]) d.weak_rec[sh.wi[g]].pmask = sh.alive[g]; }
	}
	if (DVP_LANE0 && nevals) *nevals += (unsigned long long)T;
}

// ---- D1: joint view selection and the candidates' weighted costs (APD.cu:2781-2874) -----------------------------------------
DVP_HD void weak_d1_px(const Dev& d, int px, int py, int iter) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const int wi = d.neighbours_map[center];
	const s2* nbs = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	WeakRec& rec = d.weak_rec[wi];
	const uint32_t flag = rec.flag;
	const float* ev = weak_ev_of(d, wi);
	// cost_array: `= { 2.0f }` sets one element, the rest is 0 (APD.cu:2769); rows of the anchors that count hold their costs
#define DVP_CA(k, j) (((flag >> (k)) & 1) ? e8[k] : (((k) | (j)) == 0 ? 2.0f : 0.0f))
	uint32_t nsel[8];
	bool nvalid[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const s2 nb = nbs[i + 1];
		nvalid[i] = !(nb.x == -1 || nb.y == -1);
		nsel[i] = nvalid[i] ? d.selected_views[nb.x + nb.y * W] : 0u;
	}
	const float thr = (float)(0.8 * dvp_expf((iter) * (iter) / (-90.0f)));
	float probs[32];
	for (int j = 0; j < S; ++j) {
		float pr = 0.0f;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (!nvalid[i]) continue;
			pr += is_set(nsel[i], j) ? 0.9f : 0.1f;
		}
		float count = 0;
		int count_false = 0;
		float tmpw = 0;
		float e8[8];
		weak_ev_load8(ev, j, e8);
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const float cst = DVP_CA(k, j);
			if (cst < thr) { tmpw += dvp_expf(cst * cst / (-0.18f)); count++; }
			if (cst > 1.2f) count_false++;
		}
		float p = 0.0f;
		if (count > 2 && count_false < 3) p = tmpw / count;
		else if (count_false < 3) p = dvp_expf(thr * thr / (-0.32f));
		probs[j] = p * pr;
	}
	float psum = 0.0f;
	for (int i = 0; i < S; ++i) psum += probs[i];
	const float inv = 1.0f / psum;
	float cum = 0.0f;
	for (int i = 0; i < S; ++i) {
		cum += probs[i] * inv;
		probs[i] = cum;   // the CDF
	}
	uint8_t vw[32];
	for (int i = 0; i < 32; ++i) vw[i] = 0;
	Rng rv(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_VIEW));
	for (int s = 0; s < 15; ++s) {
		const float rp = rv.uniform() - FLT_EPSILON;
		for (int v = 0; v < S; ++v)
			if (probs[v] > rp) { vw[v] += 1; break; }
	}
	uint32_t sel_mask = 0;
	float weight_norm = 0;
	for (int i = 0; i < S; ++i)
		if (vw[i] > 0) { set_bit(&sel_mask, i); weight_norm += vw[i]; }
	{
		uint32_t* out = reinterpret_cast<uint32_t*>(d.view_weight + (size_t)center * 32);
		for (int i = 0; i < 8; ++i) out[i] = (uint32_t)vw[4 * i] | ((uint32_t)vw[4 * i + 1] << 8) | ((uint32_t)vw[4 * i + 2] << 16) | ((uint32_t)vw[4 * i + 3] << 24);
	}
	// weighted candidate costs (APD.cu:2852-2874): view outside, candidate inside — every candidate's sum still runs over the
	// views in order
	float fc[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) fc[k] = 0.0f;
	for (int j = 0; j < S; ++j) {
		const int w = vw[j];
		if (w <= 0) continue;
		float e8[8];
		weak_ev_load8(ev, j, e8);
		if (P.geom_consistency) {
			const DvpCamera sc = load_camera(d, j + 1);
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				if ((flag >> k) & 1) fc[k] += w * (DVP_CA(k, j) + P.geom_factor * geom_cost_cams(d, rc, sc, j + 1, px, py, rec.pl[k]));
				else fc[k] += w * (DVP_CA(k, j) + P.geom_factor * 3.0f);
			}
		} else {
#pragma unroll
			for (int k = 0; k < 8; ++k) fc[k] += w * DVP_CA(k, j);
		}
	}
#undef DVP_CA
	int min_cost_idx = 0;
	float mc = fc[0] / weight_norm;
#pragma unroll
	for (int k = 1; k < 8; ++k) {
		const float f = fc[k] / weight_norm;
		if (f <= mc) { mc = f; min_cost_idx = k; }
	}
	const f4 fp = d.fit_planes[center];
	const bool skip_refine = fp.x == 0 && fp.y == 0 && fp.z == 0;
	f4 cand = mk4(0, 0, 0, 0);
#pragma unroll
	for (int k = 0; k < 8; ++k)
		if (k == min_cost_idx) cand = rec.pl[k];
	rec.cand = cand;
	rec.pl[0] = d.planes[center];
	if (!skip_refine) rec.pl[1] = fp;
	rec.pmask = skip_refine ? 1u : 3u;
	rec.vmask = sel_mask;
	rec.sel_mask = sel_mask;
	rec.weight_norm = weight_norm;
	rec.min_cost_idx = min_cost_idx;
	rec.fcost_min = mc;
	rec.skip_refine = skip_refine ? 1u : 0u;
}

// weighted cost of plane `pl` over the selected views (APD.cu:2876-2890 and the like): sum_j w_j (ev_j [+ factor geom_j]) / norm
// the 32 view weights of a pixel: two 16-byte loads
struct ViewWeights { uint32_t w[8]; DVP_HD int at(int j) const { return (int)((w[j >> 2] >> (8 * (j & 3))) & 255u); } };
DVP_HD ViewWeights load_view_weights(const Dev& d, int center) {
	ViewWeights r;
	const uint32_t* g32 = reinterpret_cast<const uint32_t*>(d.view_weight + (size_t)center * 32);
#pragma unroll
	for (int i = 0; i < 8; ++i) r.w[i] = g32[i];
	return r;
}
DVP_HD float weak_weighted_cost(const Dev& d, const DvpCamera& rc, int px, int py, const ViewWeights& vw, const float* ev, int q, const f4 pl, float weight_norm) {
	const DvpParams& P = d.params;
	const int S = P.num_images - 1;
	float tc = 0.0f;
	for (int j = 0; j < S; ++j) {
		const int w = vw.at(j);
		if (w > 0) {
			if (P.geom_consistency) tc += w * (ev[weak_ev_index(q, j)] + P.geom_factor * geom_cost_cams(d, rc, load_camera(d, j + 1), j + 1, px, py, pl));
			else tc += w * ev[weak_ev_index(q, j)];
		}
	}
	return tc / weight_norm;
}

// ---- D2: the current plane's cost, candidate / fit-plane adoption, the refinement hypotheses (APD.cu:2876-2960, 1897-1960) ----
DVP_HD void weak_d2_px(const Dev& d, int px, int py, int iter) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const int wi = d.neighbours_map[center];
	WeakRec& rec = d.weak_rec[wi];
	const float* ev = weak_ev_of(d, wi);
	const ViewWeights vw = load_view_weights(d, center);
	const float weight_norm = rec.weight_norm;
	const uint32_t sel_mask = rec.sel_mask;
	const bool skip_refine = rec.skip_refine != 0;
	uint32_t sel_now = d.selected_views[center];   // what random_normal_yzl reads (updated on adoption)
	const f4 pl0 = rec.pl[0];
	float cost_now = weak_weighted_cost(d, rc, px, py, vw, ev, 0, pl0, weight_norm);
	const float costs_center = cost_now;
	f4 plane_now = pl0;
	float depth_now = depth_from_plane(rc, plane_now, px, py);
	if ((rec.flag >> rec.min_cost_idx) & 1) {
		const f4 cand = rec.cand;
		const float db = depth_from_plane(rc, cand, px, py);
		if (db >= P.depth_min && db <= P.depth_max && rec.fcost_min < cost_now) {
			depth_now = db;
			plane_now = cand;
			cost_now = rec.fcost_min;
			sel_now = sel_mask;
			d.selected_views[center] = sel_mask;
		}
	}
	uint32_t keep = 0;
	if (!skip_refine) {   // fit-plane test, then the five hypotheses
		const f4 pl1 = rec.pl[1];
		const float tc = weak_weighted_cost(d, rc, px, py, vw, ev, 1, pl1, weight_norm);
		const float db = depth_from_plane(rc, pl1, px, py);
		if (db >= P.depth_min && db <= P.depth_max && tc < cost_now) {
			depth_now = db;
			plane_now = pl1;
			cost_now = tc;
		}
		Rng rd(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_DEPTH_RAND));
		Rng rn(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_NORMAL));
		Rng rp(d.seed, (uint32_t)center, rng_site(PH_WEAK, iter, SUB_DEPTH_PERT));
		const float depth_rand = rd.uniform() * (P.depth_max - P.depth_min) + P.depth_min;
		const f4 n_rand = random_normal_yzl_sel(d, px, py, rn, depth_now, sel_now);
		const float dmin_p = (1 - 0.02f) * depth_now, dmax_p = (1 + 0.02f) * depth_now;
		const float depth_pert = rp.uniform() * (dmax_p - dmin_p) + dmin_p;
		f4 n_pert = plane_now;
		normalize3(&n_pert);
		const float rdep[5] = { depth_rand, depth_now, depth_rand, depth_now, depth_pert };
		const f4 rnrm[5] = { plane_now, n_rand, n_rand, n_pert, plane_now };
#pragma unroll
		for (int i = 0; i < 5; ++i) {
			f4 h = rnrm[i];
			h.w = distance_to_origin(rc, px, py, rdep[i], h);
			rec.pl[i] = h;
			// (i) of the two exact short cuts of weak_update_wave: a hypothesis out of the depth range cannot be adopted
			const float db = depth_from_plane(rc, h, px, py);
			if (db >= P.depth_min && db <= P.depth_max) keep |= 1u << i;
		}
	}
	rec.pmask = keep;
	rec.vmask = sel_mask;
	rec.cost_now = cost_now;
	rec.costs_center = costs_center;
	rec.depth_now = depth_now;
	rec.plane_now = plane_now;
	rec.sel_now = sel_now;
}

// ---- D3: adoption of the hypotheses, the final plane (APD.cu:1361-1383, 3060-3070) ------------------------------------------
DVP_HD void weak_d3_px(const Dev& d, int px, int py) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const DvpCamera rc = load_camera(d, 0);
	const int S = P.num_images - 1;
	const int wi = d.neighbours_map[center];
	const WeakRec& rec = d.weak_rec[wi];
	const float* ev = weak_ev_of(d, wi);
	const ViewWeights vw = load_view_weights(d, center);
	const float weight_norm = rec.weight_norm;
	float cost_now = rec.cost_now, depth_now = rec.depth_now;
	f4 plane_now = rec.plane_now;
	const uint32_t pmask = rec.pmask;   // the hypotheses that are still candidates
	for (int i = 0; i < 5; ++i) {
		if (!((pmask >> i) & 1)) continue;
		const f4 h = rec.pl[i];
		const float tc = weak_weighted_cost(d, rc, px, py, vw, ev, i, h, weight_norm);
		const float db = depth_from_plane(rc, h, px, py);
		if (db >= P.depth_min && db <= P.depth_max && tc < cost_now) {
			depth_now = db;
			plane_now = h;
			cost_now = tc;
		}
	}
	f4 final_plane = d.planes[center];
	if (P.state == DVP_REFINE_INIT) {
		if (cost_now < rec.costs_center - 0.1) final_plane = plane_now;
	} else {
		final_plane = plane_now;
	}
	d.planes[center] = final_plane;
}

// ---- E3: cost of the final plane with the plain bilateral NCC at the default radius (APD.cu:3072-3088), one lane per pixel ----
template <int SMP>
DVP_HD void weak_final_cost_px(const Dev& d, int px, int py, PatchTab tab, unsigned long long* nevals) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const int S = P.num_images - 1;
	const int wi = d.neighbours_map[center];
	const ViewWeights vw = load_view_weights(d, center);
	const f4 final_plane = d.planes[center];
	PatchCtx c2;
	{
		int r = P.strong_radius, inc = P.strong_increment;
		if (P.use_radius) inc = DVP_MAX(2, (int)(2.0 * r / 5.0));
		build_patch_ctx(d, px, py, r, inc, 0, tab, &c2);
	}
	float cn = 0.0f;
	unsigned long long evals = 0;
	for (int v = 0; v < S; ++v) {
		if (vw.at(v) == 0) continue;
		cn += (uint8_t)vw.at(v) * ncc_old<SMP>(d, c2, px, py, v + 1, final_plane);
		evals += 1;
	}
	d.costs[center] = cn / d.weak_rec[wi].weight_norm;
	if (nevals) *nevals += evals;
}

}  // namespace dvp
#endif
