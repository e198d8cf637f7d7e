// dvp_weak_phased.hpp — the weak-pixel update (CheckerboardPropagationWeak + PlaneHypothesisRefinementWeak,
// APD.cu:2739-3089, 1897-2008) as SEVEN launches: evaluation launches with one wave per WEAK pixel and decision launches
// with one LANE per WEAK pixel.
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
//   E2  wave   hypotheses in range against the first selected view, the survivors against the rest -> ev
//   D3  lane   adoption of the hypotheses, the final plane
//   E3  lane   cost of the final plane with the plain bilateral NCC (the per-lane evaluator of the strong path)
//
// Between launches a pixel's state travels in a 256-byte record (WeakRec), its centre-patch table (36 x (w, w ref)) and
// its cost vectors (8 x S floats) — per WEAK pixel, indexed like Dev::neighbours.  Every floating-point operation is the
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

DVP_HD float* weak_ev_of(const Dev& d, int wi) { return d.weak_ev + (size_t)wi * 8 * (size_t)(d.params.num_images - 1); }

// the centre-patch context the update's first launch left (record + table) -> sh.ctab, c
template <class SH>
DVP_HD void weak_load_ctx(const Dev& d, int wi, SH& sh, PatchCtx* c) {
	const WeakRec& rec = d.weak_rec[wi];
	c->tab = PatchTab{nullptr, 0};
	c->radius = rec.radius;
	c->inc = rec.inc;
	c->fast = rec.fast;
	c->sum_ref = rec.sum_ref;
	c->sum_ref_ref = rec.sum_ref_ref;
	c->wsum = rec.wsum;
	DVP_LANES(t) {
		if (t < kTaps * kTaps) sh.ctab[t] = d.weak_ctab[(size_t)wi * (kTaps * kTaps) + t];
		if (t < 8) sh.pl[t] = rec.pl[t];
	}
	wave_sync();
}
// sh.ev[q][v] for q in pmask, v in vmask -> the pixel's cost vectors
template <class SH>
DVP_HD void weak_store_ev(const Dev& d, int wi, uint32_t pmask, uint32_t vmask, const SH& sh) {
	const int S = d.params.num_images - 1;
	float* ev = weak_ev_of(d, wi);
	DVP_LANES(l) {
		for (int i = l; i < 8 * S; i += 64) {
			const int q = i / S, v = i - q * S;
			if (((pmask >> q) & 1) && ((vmask >> v) & 1)) ev[i] = sh.ev[q][v];
		}
	}
}

// ---- E0: the propagation candidates ---------------------------------------------------------------------------------------
template <int SMP, int FMT, int TAB>
DVP_HD void weak_e0_wave(const Dev& d, int px, int py, unsigned long long* nevals, WeakSharedT<TAB>& sh) {
	const int W = d.width;
	const int center = py * W + px;
	const int S = d.params.num_images - 1;
	const uint32_t all_views = (S >= 32) ? 0xFFFFFFFFu : ((1u << S) - 1u);
	const int wi = d.neighbours_map[center];
	const s2* nbs = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	const float cpix = ref_texel_t<FMT>(d, px, py);
	PatchCtx c;
	c.tab = PatchTab{nullptr, 0};
	{
		int radius, inc;
		patch_geometry(d, center, &radius, &inc);
		wave_patch_ctx<FMT>(d, px, py, radius, inc, 1, sh, &c);
	}
	// the anchors' planes (APD.cu:2771-2779): lane k owns anchor k
	uint32_t flag = 0;
	DVP_LANES(k) {
		bool ok = false;
		if (k < 8) {
			const s2 nb = nbs[k + 1];
			if (!(nb.x == -1 || nb.y == -1) && d.weak_info[nb.x + nb.y * W] == DVP_STRONG) {
				ok = true;
				sh.pl[k] = d.planes[nb.x + nb.y * W];
			}
		}
		flag |= wave_lane_bit(ok, k);
	}
	wave_sync();
	if (flag) {
		weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, all_views, flag, sh);
		weak_store_ev(d, wi, flag, all_views, sh);
	}
	WeakRec& rec = d.weak_rec[wi];
	DVP_LANES(t) {
		if (t < kTaps * kTaps) d.weak_ctab[(size_t)wi * (kTaps * kTaps) + t] = sh.ctab[t];
		if (t < 8 && ((flag >> t) & 1)) rec.pl[t] = sh.pl[t];
	}
	if (DVP_LANE0) {
		rec.flag = flag;
		rec.radius = c.radius; rec.inc = c.inc; rec.fast = c.fast;
		rec.sum_ref = c.sum_ref; rec.sum_ref_ref = c.sum_ref_ref; rec.wsum = c.wsum;
		if (nevals) *nevals += (unsigned long long)__builtin_popcount(flag) * (unsigned long long)__builtin_popcount(all_views);
	}
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
#define DVP_CA(k, j) (((flag >> (k)) & 1) ? ev[(k) * S + (j)] : (((k) | (j)) == 0 ? 2.0f : 0.0f))
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

// ---- E1: the planes of the record against its views ----------------------------------------------------------------------
template <int SMP, int FMT, int TAB>
DVP_HD void weak_e1_wave(const Dev& d, int px, int py, unsigned long long* nevals, WeakSharedT<TAB>& sh) {
	const int W = d.width;
	const int center = py * W + px;
	const int wi = d.neighbours_map[center];
	const s2* nbs = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	const WeakRec& rec = d.weak_rec[wi];
	const uint32_t pmask = rec.pmask, vmask = rec.vmask;
	if (!(pmask && vmask)) return;
	const float cpix = ref_texel_t<FMT>(d, px, py);
	PatchCtx c;
	weak_load_ctx(d, wi, sh, &c);
	weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, vmask, pmask, sh);
	weak_store_ev(d, wi, pmask, vmask, sh);
	if (DVP_LANE0 && nevals) *nevals += (unsigned long long)__builtin_popcount(pmask) * (unsigned long long)__builtin_popcount(vmask);
}

// weighted cost of plane `pl` over the selected views (APD.cu:2876-2890 and the like): sum_j w_j (ev_j [+ factor geom_j]) / norm
DVP_HD float weak_weighted_cost(const Dev& d, const DvpCamera& rc, int px, int py, const uint8_t* vw, const float* evq, const f4 pl, float weight_norm) {
	const DvpParams& P = d.params;
	const int S = P.num_images - 1;
	float tc = 0.0f;
	for (int j = 0; j < S; ++j) {
		const int w = vw[j];
		if (w > 0) {
			if (P.geom_consistency) tc += w * (evq[j] + P.geom_factor * geom_cost_cams(d, rc, load_camera(d, j + 1), j + 1, px, py, pl));
			else tc += w * evq[j];
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
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	const float weight_norm = rec.weight_norm;
	const uint32_t sel_mask = rec.sel_mask;
	const bool skip_refine = rec.skip_refine != 0;
	uint32_t sel_now = d.selected_views[center];   // what random_normal_yzl reads (updated on adoption)
	const f4 pl0 = rec.pl[0];
	float cost_now = weak_weighted_cost(d, rc, px, py, vw, ev, pl0, weight_norm);
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
		const float tc = weak_weighted_cost(d, rc, px, py, vw, ev + S, pl1, weight_norm);
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

// ---- E2: the hypotheses against the first selected view, the survivors against the rest -------------------------------------
template <int SMP, int FMT, int TAB>
DVP_HD void weak_e2_wave(const Dev& d, int px, int py, unsigned long long* nevals, WeakSharedT<TAB>& sh) {
	const int W = d.width;
	const int center = py * W + px;
	const DvpParams& P = d.params;
	const int wi = d.neighbours_map[center];
	const s2* nbs = d.neighbours + (size_t)wi * DVP_NEIGHBOUR_NUM;
	WeakRec& rec = d.weak_rec[wi];
	const uint32_t pmask = rec.pmask, vmask = rec.vmask;
	if (!(pmask && vmask)) return;
	const float cpix = ref_texel_t<FMT>(d, px, py);
	const float cost_now = rec.cost_now, weight_norm = rec.weight_norm;
	PatchCtx c;
	weak_load_ctx(d, wi, sh, &c);
	// (ii): the weighted sum only grows (weights > 0, costs >= 0, IEEE addition and division are monotone), so a hypothesis whose
	// FIRST selected view alone is not below the best cost at entry can never be adopted
	const int first = __builtin_ctz(vmask);
	weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, 1u << first, pmask, sh);
	unsigned long long evals = (unsigned long long)__builtin_popcount(pmask);
	const int w = d.view_weight[(size_t)center * 32 + first];
	const DvpCamera rc = load_camera(d, 0), fc = load_camera(d, first + 1);
	uint32_t alive = 0;
	DVP_LANES(i) {
		bool ok = false;
		if (i < 5 && ((pmask >> i) & 1)) {
			float tc = 0.0f;
			if (P.geom_consistency) tc += w * (sh.ev[i][first] + P.geom_factor * geom_cost_cams(d, rc, fc, first + 1, px, py, sh.pl[i]));
			else tc += w * sh.ev[i][first];
			ok = tc / weight_norm < cost_now;
		}
		alive |= wave_lane_bit(ok, i);
	}
	const uint32_t rest = vmask & ~(1u << first);
	if (alive && rest) {
		weak_eval<SMP, FMT>(d, c, nbs, cpix, px, py, rest, alive, sh);
		evals += (unsigned long long)__builtin_popcount(alive) * (unsigned long long)__builtin_popcount(rest);
	}
	weak_store_ev(d, wi, alive, vmask, sh);
	if (DVP_LANE0) {
		rec.pmask = alive;
		if (nevals) *nevals += evals;
	}
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
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
	const float weight_norm = rec.weight_norm;
	float cost_now = rec.cost_now, depth_now = rec.depth_now;
	f4 plane_now = rec.plane_now;
	const uint32_t pmask = rec.pmask;   // the hypotheses that are still candidates
	for (int i = 0; i < 5; ++i) {
		if (!((pmask >> i) & 1)) continue;
		const f4 h = rec.pl[i];
		const float tc = weak_weighted_cost(d, rc, px, py, vw, ev + i * S, h, weight_norm);
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
	const uint8_t* vw = d.view_weight + (size_t)center * 32;
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
		if (vw[v] == 0) continue;
		cn += vw[v] * ncc_old<SMP>(d, c2, px, py, v + 1, final_plane);
		evals += 1;
	}
	d.costs[center] = cn / d.weak_rec[wi].weight_norm;
	if (nevals) *nevals += evals;
}

}  // namespace dvp
#endif
