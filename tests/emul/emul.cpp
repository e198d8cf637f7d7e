// TEST INFRASTRUCTURE — host emulation of the engine's kernels.
//
// Compiles the SAME device headers the HIP engine is built from (dvp-mvs_amd/csrc/*.hpp) with
// g++ and emulates kernel launches by looping over (block, wave, lane) through the same
// block->pixel map.  Purpose: check kernel logic, launch geometry and bit-parity against the CPU
// oracle on machines without a GPU (`pytest -m "not gpu"`).  It is built only by the test suite
// into tests/emul/; libdvp_mvs_hip.so neither contains nor falls back to it.
#include "../../dvp-mvs_amd/csrc/dvp_stages.hpp"
#include <cstring>
#include <cstdio>
#include <vector>
#include <cstdlib>

using namespace dvp;

namespace {

struct Emu {
	int W, H, NI, pitch;
	std::vector<float> images, depths;
	std::vector<uint8_t> images8;       // byte row pairs (Dev::images8), valid when every image set so far is 8-bit exact
	std::vector<char> image_exact;
	std::vector<DvpCamera> cameras;
	std::vector<ViewConst> views;
	std::vector<int> sector_taps, sector_start;
	std::vector<f4> planes, planes_snap, fit_planes;
	std::vector<int> search_pos;
	std::vector<float> slot_costs, strong_rec;   // the split strong update's hand-over buffers (dvp_strong.hpp)
	std::vector<f4> sweep_rec;                   // DepthToWeak + LocalRefine as view-compacted passes (dvp_strong.hpp: sweep_*)
	std::vector<float> sweep_cost, sweep_pc;
	std::vector<float> costs, costs_snap, complex_;
	std::vector<uint32_t> selected_views;
	std::vector<uint8_t> view_weight, weak_info, weak_reliable, edge;
	std::vector<uint32_t> edge_bits, strong_bits, strong_bits_t;
	std::vector<int> edge_sat;
	std::vector<s2> weak_nearest_strong, neighbours, candidate, edge_neigh, label_boundary, label_stop, gn_points;
	std::vector<int> gn_count;
	std::vector<AnchorRec> anchor_tab;   // the weak update's per-pass table (dvp_weak_wave.hpp), same validity rule as the engine's
	bool anchor_tab_valid = false, anchor_tab_off = false;
	std::vector<WeakRec> weak_rec;       // the weak update as seven launches (dvp_weak_phased.hpp)
	std::vector<f2> weak_ctab;
	std::vector<float> weak_ev;
	std::vector<int> neighbours_map, label, radius;
	unsigned long long evals = 0;
	bool count = false;
	Dev d;
};

void refresh(Emu& e) {
	Dev& d = e.d;
	d.width = e.W; d.height = e.H; d.num_images = e.NI; d.pitch = e.pitch;
	d.org = kImgPad * e.pitch + kImgPad;
	d.plane_stride = (size_t)e.pitch * (e.H + 2 * kImgPad);
	d.images = e.images.data();
	d.img8_tiles_x = img8_tiles_x(e.W);
	d.img8_plane_bytes = (size_t)img8_tiles_x(e.W) * img8_tiles_y(e.H) * 128;
	{
		bool all = !e.image_exact.empty();
		for (char ok : e.image_exact) all = all && ok;
		d.images8 = all ? e.images8.data() : nullptr;
	}
	d.depths = e.depths.data();
	d.cameras = e.cameras.data();
	d.views = e.views.data();
	d.sector_taps = e.sector_taps.data();
	d.sector_start = e.sector_start.data();
	d.search_pos = e.search_pos.data();
	d.slot_costs = e.slot_costs.data(); d.strong_rec = e.strong_rec.data(); d.half_w = (e.W + 1) / 2;
	d.sweep_rec = e.sweep_rec.data(); d.sweep_cost = e.sweep_cost.data(); d.sweep_pc = e.sweep_pc.data();
	d.sweep_px0 = 0; d.sweep_row0 = 0; d.sweep_row1 = 0;   // (the engine's bands of rows: a launch-level matter, the whole image here)
	d.planes = e.planes.data(); d.planes_snap = e.planes_snap.data();
	d.costs = e.costs.data(); d.costs_snap = e.costs_snap.data();
	d.selected_views = e.selected_views.data();
	d.view_weight = e.view_weight.data();
	d.weak_info = e.weak_info.data();
	d.weak_reliable = e.weak_reliable.data();
	d.weak_nearest_strong = e.weak_nearest_strong.data();
	d.neighbours_map = e.neighbours_map.data();
	d.neighbours = e.neighbours.data();
	d.gn_points = e.gn_points.data();
	d.gn_count = e.gn_count.data();
	d.fit_planes = e.fit_planes.data();
	d.candidate = e.candidate.data();
	d.anchor_tab = (e.anchor_tab_off || !e.anchor_tab_valid) ? nullptr : e.anchor_tab.data();
	d.weak_rec = e.weak_rec.data(); d.weak_ctab = e.weak_ctab.data(); d.weak_ev = e.weak_ev.data();
	d.edge = e.edge.data();
	d.edge_bits = e.edge_bits.data();
	d.edge_sat = e.edge_sat.data();
	d.sat_cells_x = sat_cells(e.W);
	d.sat_cells_y = sat_cells(e.H);
	d.strong_bits = e.strong_bits.data();
	d.strong_bits_t = e.strong_bits_t.data();
	d.edge_tiles_x = edge_tiles_x(e.W);
	d.edge_neigh = e.edge_neigh.data();
	d.label = e.label.data();
	d.label_boundary = e.label_boundary.data();
	d.label_stop = e.label_stop.data();
	d.complex_ = e.complex_.data();
	d.radius = e.radius.data();
	d.eval_counter = nullptr;
}

template <int STAGE>
void launch(Emu& e, int iter, int colour) {
	const LaunchGeom g = make_geom(e.W, e.H, stage_is_half(STAGE));
	unsigned long long total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
	for (int b = 0; b < g.grid(); ++b)
		for (int wave = 0; wave < 4; ++wave)
			for (int lane = 0; lane < 64; ++lane) {
				int px, py;
				if (!block_to_pixel(b, lane, wave, g.tiles_x, g.tiles, g.rows, g.half ? 1 : 0, colour, e.W, e.H, &px, &py)) continue;
				unsigned long long n = 0;
				f2 tab_mem[kTaps * kTaps];
				const PatchTab tab{tab_mem, 1};
				const bool narrow = e.NI - 1 <= kNarrowViews;   // the engine's strong-update dispatch
				if (e.d.sampler) { if (narrow) run_pixel<STAGE, 1, kNarrowViews>(e.d, px, py, iter, e.count ? &n : nullptr, tab); else run_pixel<STAGE, 1>(e.d, px, py, iter, e.count ? &n : nullptr, tab); }
				else { if (narrow) run_pixel<STAGE, 0, kNarrowViews>(e.d, px, py, iter, e.count ? &n : nullptr, tab); else run_pixel<STAGE, 0>(e.d, px, py, iter, e.count ? &n : nullptr, tab); }
				total += n;
			}
	e.evals += total;
}

}  // namespace

extern "C" {

void* emu_create(int W, int H, int NI) {
	Emu* e = new Emu();
	e->W = W; e->H = H; e->NI = NI;
	e->pitch = (W + 2 * kImgPad + 63) / 64 * 64;
	const size_t L = (size_t)W * H;
	const int S = NI - 1;
	e->images.assign((size_t)e->pitch * (H + 2 * kImgPad) * NI * 2, 0.0f);   // row-pair planes
	e->images8.assign((size_t)img8_tiles_x(W) * img8_tiles_y(H) * 128 * NI, 0);
	e->image_exact.assign(NI, 0);
	e->depths.assign((size_t)e->pitch * (H + 2 * kImgPad) * NI, 0.0f);
	e->cameras.resize(NI);
	e->views.resize(NI);
	e->planes.assign(L, mk4(0, 0, 0, 0));
	e->planes_snap = e->planes;
	e->search_pos.assign(L * 16, -1);
	e->slot_costs.assign((size_t)kSlotCount * (S > 0 ? S : 1) * ((W + 1) / 2) * H, 0.0f);
	e->strong_rec.assign((size_t)SR_FIELDS * ((W + 1) / 2) * H, 0.0f);
	e->fit_planes = e->planes;
	e->costs.assign(L, 0.0f);
	e->costs_snap = e->costs;
	e->selected_views.assign(L + W, 0u);
	e->view_weight.assign(L * 32, 0);
	e->weak_info.assign(L, (uint8_t)DVP_STRONG);
	e->weak_reliable.assign(L, 0);
	e->weak_nearest_strong.assign(L, mks2(-1, -1));
	e->neighbours_map.assign(L, 0);
	e->neighbours.assign(DVP_NEIGHBOUR_NUM, mks2(-1, -1));
	e->gn_points.assign(kGnDirSlots, mks2(-1, -1));
	e->gn_count.assign(1, 0);
	e->candidate.assign(L * (size_t)(S > 0 ? S : 1) * 8, mks2(0, 0));
	e->edge.assign(L, 0);
	e->edge_bits.assign(edge_bits_words(W, H), 0u);
	e->edge_sat.assign((size_t)(sat_cells(W) + 1) * (sat_cells(H) + 1), 0);
	e->strong_bits.assign(edge_bits_words(W, H), 0u);
	e->strong_bits_t.assign(edge_bits_words(W, H), 0u);
	e->edge_neigh.assign(L * 8, mks2(-1, -1));
	e->label_stop.assign(L * 8, mks2(-1, -1));
	e->label.assign(L, 0);
	e->label_boundary.assign(8, mks2(-1, -1));
	e->complex_.assign(1, 0.0f);
	e->radius.assign(L, 5);
	std::memset(&e->d, 0, sizeof(Dev));
	make_sector_taps(5, &e->sector_taps, &e->sector_start);
	refresh(*e);
	return e;
}
void emu_destroy(void* c) { delete (Emu*)c; }
// padded plane with replicated borders (== clamp addressing)
static void fill_plane(Emu& e, float* plane, const float* data) {
	for (int y = -kImgPad; y < e.H + kImgPad; ++y)
		for (int x = -kImgPad; x < e.W + kImgPad; ++x)
			plane[(size_t)(y + kImgPad) * e.pitch + (x + kImgPad)] = data[(size_t)clampi(y, 0, e.H - 1) * e.W + clampi(x, 0, e.W - 1)];
}
void emu_set_image(void* c, int idx, const float* data) {
	Emu& e = *(Emu*)c;
	e.anchor_tab_valid = false;
	std::vector<float> plain(e.d.plane_stride, 0.0f);
	fill_plane(e, plain.data(), data);
	const int PH = e.H + 2 * kImgPad;
	float* out = &e.images[(size_t)idx * e.d.plane_stride * 2];   // {I(x,y), I(x,y+1)} (dvp_interleave_rows)
	for (int y = 0; y < PH; ++y)
		for (int x = 0; x < e.pitch; ++x) {
			out[((size_t)y * e.pitch + x) * 2] = plain[(size_t)y * e.pitch + x];
			out[((size_t)y * e.pitch + x) * 2 + 1] = plain[(size_t)(y + 1 < PH ? y + 1 : y) * e.pitch + x];
		}
	// tiled byte planes (dvp_pairs_to_tiles)
	const int t8x = img8_tiles_x(e.W), t8y = img8_tiles_y(e.H), PW = e.W + 2 * kImgPad;
	uint8_t* out8 = &e.images8[(size_t)idx * t8x * t8y * 128];
	bool exact = true;
	for (int ty = 0; ty < t8y && exact; ++ty)
		for (int tx = 0; tx < t8x && exact; ++tx)
			for (int el = 0; el < kT8E * kT8H; ++el) {
				const int sx = std::min(tx * kT8W + (el % kT8E), PW - 1), sy = std::min(ty * kT8H + (el / kT8E), PH - 1);
				const int sx1 = std::min(sx + 1, PW - 1);
				uint8_t* dst = &out8[(size_t)(ty * t8x + tx) * 128 + (size_t)el * kT8B];
				for (int c2 = 0; c2 < kT8B; ++c2) {
					const float v = out[((size_t)sy * e.pitch + (c2 < 2 ? sx : sx1)) * 2 + (c2 & 1)];
					if (!(v >= 0.0f && v <= 255.0f && v == floorf(v))) { exact = false; break; }
					dst[c2] = (uint8_t)v;
				}
				if (!exact) break;
			}
	e.image_exact[idx] = exact ? 1 : 0;
	refresh(e);
}
void emu_set_depth(void* c, int idx, const float* data) {
	Emu& e = *(Emu*)c;
	fill_plane(e, &e.depths[(size_t)idx * e.d.plane_stride], data);
}
void emu_set_cameras(void* c, const DvpCamera* cams, int n) {
	Emu& e = *(Emu*)c;
	for (int i = 0; i < n; ++i) e.cameras[i] = cams[i];
	for (int i = 1; i < n; ++i) compute_view_const(e.cameras[0], e.cameras[i], &e.views[i]);
}
void emu_set_params(void* c, const DvpParams* p) {
	Emu& e = *(Emu*)c;
	e.anchor_tab_valid = false;
	e.d.params = *p;
	set_neighbour_consts(&e.d);
	make_sector_taps(p->weak_radius, &e.sector_taps, &e.sector_start);
	refresh(e);
}
void emu_set_seed(void* c, uint64_t s) { ((Emu*)c)->d.seed = s; }
void emu_set_sampler(void* c, int s) { ((Emu*)c)->d.sampler = s; }
void emu_count_evals(void* c, int on) { Emu& e = *(Emu*)c; e.count = on != 0; e.evals = 0; }
long long emu_get_evals(void* c) { return (long long)((Emu*)c)->evals; }

void emu_upload_state(void* c, const f4* planes, const uint32_t* views, const uint8_t* weak,
	const uint8_t* edge, const int* label, const int* radius) {
	Emu& e = *(Emu*)c;
	e.anchor_tab_valid = false;
	const size_t L = (size_t)e.W * e.H;
	if (planes) std::memcpy(e.planes.data(), planes, L * sizeof(f4));
	if (views) std::memcpy(e.selected_views.data(), views, L * 4);
	if (edge) std::memcpy(e.edge.data(), edge, L);
	if (label) std::memcpy(e.label.data(), label, L * 4);
	if (radius) std::memcpy(e.radius.data(), radius, L * 4);
	if (weak) std::memcpy(e.weak_info.data(), weak, L);
	int wc = 0;
	for (size_t i = 0; i < L; ++i) {
		e.neighbours_map[i] = 0;
		if (e.weak_info[i] == DVP_WEAK) e.neighbours_map[i] = wc++;
	}
	e.d.weak_count = wc;
	const size_t n = (size_t)(wc > 0 ? wc : 1);
	e.neighbours.assign(n * DVP_NEIGHBOUR_NUM, mks2(-1, -1));
	e.gn_points.assign(n * kGnDirSlots, mks2(-1, -1));
	e.gn_count.assign(n, 0);
	e.complex_.assign(n, 0.0f);
	e.label_boundary.assign(n * 8, mks2(-1, -1));
	refresh(e);
}

static void* buf_ptr(Emu& e, int id, size_t* bytes) {
	const size_t L = (size_t)e.W * e.H;
	switch (id) {
	case DVP_BUF_PLANES: *bytes = L * 16; return e.planes.data();
	case DVP_BUF_COSTS: *bytes = L * 4; return e.costs.data();
	case DVP_BUF_SELECTED_VIEWS: *bytes = L * 4; return e.selected_views.data();
	case DVP_BUF_VIEW_WEIGHT: *bytes = L * 32; return e.view_weight.data();
	case DVP_BUF_WEAK_INFO: *bytes = L; return e.weak_info.data();
	case DVP_BUF_WEAK_RELIABLE: *bytes = L; return e.weak_reliable.data();
	case DVP_BUF_WEAK_NEAREST_STRONG: *bytes = L * 4; return e.weak_nearest_strong.data();
	case DVP_BUF_NEIGHBOURS_MAP: *bytes = L * 4; return e.neighbours_map.data();
	case DVP_BUF_NEIGHBOURS: *bytes = e.neighbours.size() * 4; return e.neighbours.data();
	case DVP_BUF_FIT_PLANES: *bytes = L * 16; return e.fit_planes.data();
	case DVP_BUF_CANDIDATE: *bytes = e.candidate.size() * 4; return e.candidate.data();
	case DVP_BUF_EDGE: *bytes = L; return e.edge.data();
	case DVP_BUF_EDGE_NEIGH: *bytes = L * 32; return e.edge_neigh.data();
	case DVP_BUF_LABEL: *bytes = L * 4; return e.label.data();
	case DVP_BUF_LABEL_BOUNDARY: *bytes = e.label_boundary.size() * 4; return e.label_boundary.data();
	case DVP_BUF_COMPLEX: *bytes = e.complex_.size() * 4; return e.complex_.data();
	case DVP_BUF_RADIUS: *bytes = L * 4; return e.radius.data();
	}
	*bytes = 0;
	return nullptr;
}
long long emu_buffer_bytes(void* c, int id) { size_t b; buf_ptr(*(Emu*)c, id, &b); return (long long)b; }
// the candidate buffer is [view][pixel][8] inside the kernels, [pixel][view][8] at the boundary
static void cand_transpose(Emu& e, const s2* src, s2* dst, bool to_dev) {
	const size_t L = (size_t)e.W * e.H, S = (size_t)e.NI - 1;
	for (size_t v = 0; v < S; ++v)
		for (size_t p = 0; p < L; ++p)
			std::memcpy(to_dev ? dst + (v * L + p) * 8 : dst + (p * S + v) * 8, to_dev ? src + (p * S + v) * 8 : src + (v * L + p) * 8, 8 * sizeof(s2));
}
int emu_get_buffer(void* c, int id, void* dst) {
	size_t b; void* p = buf_ptr(*(Emu*)c, id, &b); if (!p) return -1;
	if (id == DVP_BUF_CANDIDATE) cand_transpose(*(Emu*)c, (const s2*)p, (s2*)dst, false); else std::memcpy(dst, p, b);
	return 0;
}
int emu_set_buffer(void* c, int id, const void* src) {
	((Emu*)c)->anchor_tab_valid = false;
	size_t b; void* p = buf_ptr(*(Emu*)c, id, &b); if (!p) return -1;
	if (id == DVP_BUF_CANDIDATE) cand_transpose(*(Emu*)c, (const s2*)src, (s2*)p, true); else std::memcpy(p, src, b);
	return 0;
}
int emu_weak_count(void* c) { return ((Emu*)c)->d.weak_count; }

// Launch-geometry property: the block -> tile -> pixel map of a launch visits every pixel it is meant
// to cover exactly once.  Returns 0 when it does; otherwise 1 + the first offending pixel index.
long long emu_tile_map_check(int W, int H, int half, int colour) {
	const LaunchGeom g = make_geom(W, H, half != 0);
	std::vector<int> hits((size_t)W * H, 0);
	for (int b = 0; b < g.grid(); ++b)
		for (int wave = 0; wave < 4; ++wave)
			for (int lane = 0; lane < 64; ++lane) {
				int px, py;
				if (block_to_pixel(b, lane, wave, g.tiles_x, g.tiles, g.rows, half, colour, W, H, &px, &py)) {
					if (px < 0 || py < 0 || px >= W || py >= H) return -1;
					hits[(size_t)py * W + px]++;
				}
			}
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x) {
			int want = 1;
			if (half) want = (((x & 1) ^ colour) == (y & 1)) && (y / 2) < g.rows ? 1 : 0;   // y = 2*pair + ((x&1)^colour)
			if (hits[(size_t)y * W + x] != want) return 1 + (long long)y * W + x;
		}
	return 0;
}

// dvp_pack_edge_bits
static void pack_edge(Emu& e) {
	for (size_t w = 0; w < e.edge_bits.size(); ++w) e.edge_bits[w] = pack_edge_word(e.edge.data(), e.W, e.H, edge_tiles_x(e.W), w);
	// the cell table (dvp_edge_cell_counts + the two prefix passes)
	const int CX = sat_cells(e.W), CY = sat_cells(e.H), P = CX + 1;
	std::fill(e.edge_sat.begin(), e.edge_sat.end(), 0);
	for (int y = 0; y < e.H; ++y)
		for (int x = 0; x < e.W; ++x)
			if (e.edge[(size_t)y * e.W + x]) e.edge_sat[(size_t)((y >> 3) + 1) * P + (x >> 3) + 1]++;
	for (int r = 1; r <= CY; ++r)
		for (int x = 1; x <= CX; ++x) e.edge_sat[(size_t)r * P + x] += e.edge_sat[(size_t)r * P + x - 1];
	for (int r = 1; r <= CY; ++r)
		for (int x = 0; x <= CX; ++x) e.edge_sat[(size_t)r * P + x] += e.edge_sat[(size_t)(r - 1) * P + x];
}

int emu_run_stage(void* c, int stage, int iter, int colour) {
	Emu& e = *(Emu*)c;
	// the engine's rule (dvp_engine.hip: launch_stage / ensure_anchor_table): any launch other than the three of the iteration
	// loop may change anchors, offsets or WEAK states; the first weak update after it rebuilds the table
	if (stage != DVP_ST_STRONG_UPDATE && stage != DVP_ST_RANSAC_FIT && stage != DVP_ST_WEAK_UPDATE) e.anchor_tab_valid = false;
	if (stage == DVP_ST_WEAK_UPDATE) {
		const char* off = getenv("DVP_WEAK_ANCHOR_TAB");
		e.anchor_tab_off = off && atoi(off) == 0;
		if (!e.anchor_tab_off && !e.anchor_tab_valid) {
			const int S = e.NI - 1;
			e.anchor_tab.assign((size_t)std::max(e.d.weak_count, 1) * S * kAnchors, AnchorRec{});
			e.anchor_tab_valid = true;
			refresh(e);
			const long long L = (long long)e.W * e.H;
#pragma omp parallel for schedule(dynamic, 64)
			for (long long center = 0; center < L; ++center)
				if (e.weak_info[(size_t)center] == DVP_WEAK)
					for (int v0 = 0; v0 < S; ++v0)
						for (int k = 0; k < kAnchors; ++k) build_anchor_record(e.d, (int)center, v0, k);
		} else refresh(e);
	}
	switch (stage) {
	case DVP_ST_GEN_EDGE_INFORM:
		if (e.d.params.use_edge) {
#pragma omp parallel for schedule(dynamic, 16) collapse(2)
			for (int k = 0; k < 8; ++k)
				for (int line = 0; line < e.W + e.H; ++line) edge_ray_line(e.d, k, line);
		}
		if (e.d.params.use_label && e.d.weak_count > 0) {
#pragma omp parallel for schedule(dynamic, 16) collapse(2)
			for (int k = 0; k < 8; ++k)
				for (int line = 0; line < e.W + e.H; ++line) edge_ray_line(e.d, k, line, 1);
		}
		launch<DVP_ST_GEN_EDGE_INFORM>(e, iter, colour);
		break;
	case DVP_ST_FIND_NEAREST_STRONG:
		for (size_t w = 0; w < e.strong_bits.size(); ++w) {
			e.strong_bits[w] = pack_edge_word(e.weak_info.data(), e.W, e.H, edge_tiles_x(e.W), w, (int)DVP_STRONG);
			e.strong_bits_t[w] = pack_edge_word_t(e.weak_info.data(), e.W, e.H, edge_tiles_x(e.W), w, (int)DVP_STRONG);
		}
		launch<DVP_ST_FIND_NEAREST_STRONG>(e, iter, colour);
		break;
	case DVP_ST_GEN_NEIGHBOURS:
		pack_edge(e);
		for (size_t w = 0; w < e.strong_bits.size(); ++w) e.strong_bits[w] = pack_edge_word(e.weak_info.data(), e.W, e.H, edge_tiles_x(e.W), w, (int)DVP_STRONG);
		launch<DVP_ST_GEN_NEIGHBOURS>(e, iter, colour);
		break;
	case DVP_ST_NEIGHBOUR_UPDATE: launch<DVP_ST_NEIGHBOUR_UPDATE>(e, iter, colour); break;
	case DVP_ST_RANDOM_INIT: launch<DVP_ST_RANDOM_INIT>(e, iter, colour); break;
	case DVP_ST_STRONG_UPDATE: {
		e.planes_snap = e.planes;
		e.costs_snap = e.costs;
		refresh(e);
		// dvp_strong_search: same red/black geometry
		const LaunchGeom g = make_geom(e.W, e.H, true);
#pragma omp parallel for schedule(dynamic, 1)
		for (int b = 0; b < g.grid(); ++b)
			for (int wave = 0; wave < 4; ++wave)
				for (int lane = 0; lane < 64; ++lane) {
					int px, py;
					if (block_to_pixel(b, lane, wave, g.tiles_x, g.tiles, g.rows, 1, colour, e.W, e.H, &px, &py)) strong_search_px(e.d, px, py);
				}
		// the engine issues the update as three launches for S <= 16 (dvp_strong_eval / _decide / _refine) unless
		// DVP_STRONG_SPLIT=0; the emulation follows the same switch, so both forms are checked against the oracle
		const char* sp = getenv("DVP_STRONG_SPLIT");
		const char* rl = getenv("DVP_REFINE_LANES");
		const bool refine_lanes = !(rl && atoi(rl) == 0);   // dvp_strong_refine_lanes / dvp_strong_refine
		const int S = e.NI - 1;
		if (S <= 16 && !(sp && atoi(sp) == 0)) {
			unsigned long long total = 0;
			for (int part = 0; part < 3; ++part) {
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
				for (int b = 0; b < g.grid(); ++b)
					for (int wave = 0; wave < 4; ++wave)
						for (int lane = 0; lane < 64; ++lane) {
							int px, py;
							if (!block_to_pixel(b, lane, wave, g.tiles_x, g.tiles, g.rows, 1, colour, e.W, e.H, &px, &py)) continue;
							if (e.d.weak_info[px + py * e.W] == DVP_WEAK) continue;
							unsigned long long n = 0;
							f2 tab_mem[kTaps * kTaps];
							const PatchTab tab{tab_mem, 1};
							if (part == 0) { if (e.d.sampler) strong_eval_px<1>(e.d, px, py, tab, e.count ? &n : nullptr); else strong_eval_px<0>(e.d, px, py, tab, e.count ? &n : nullptr); }
							else if (part == 1) {
								if (S <= 4) strong_decide_px<4>(e.d, px, py, iter);
								else if (S <= 6) strong_decide_px<6>(e.d, px, py, iter);
								else if (S <= 8) strong_decide_px<8>(e.d, px, py, iter);
								else if (S <= 10) strong_decide_px<10>(e.d, px, py, iter);
								else if (S <= 12) strong_decide_px<12>(e.d, px, py, iter);
								else strong_decide_px<16>(e.d, px, py, iter);
							}
							else if (refine_lanes) { if (e.d.sampler) strong_refine_px<1, true>(e.d, px, py, tab, e.count ? &n : nullptr); else strong_refine_px<0, true>(e.d, px, py, tab, e.count ? &n : nullptr); }
							else { if (e.d.sampler) strong_refine_px<1>(e.d, px, py, tab, e.count ? &n : nullptr); else strong_refine_px<0>(e.d, px, py, tab, e.count ? &n : nullptr); }
							total += n;
						}
			}
			e.evals += total;
		} else {
			launch<DVP_ST_STRONG_UPDATE>(e, iter, colour);
		}
		break;
	}
	case DVP_ST_RANSAC_FIT: pack_edge(e); launch<DVP_ST_RANSAC_FIT>(e, iter, colour); break;
	case DVP_ST_WEAK_UPDATE: {
		// the engine issues the update as seven launches (dvp_weak_phased.hpp) where the anchor table is on, unless
		// DVP_WEAK_PHASED=0; the emulation follows the same switch, so both forms are checked against the oracle
		const char* ph = getenv("DVP_WEAK_PHASED");
		if (!e.d.anchor_tab || (ph && atoi(ph) == 0)) { launch<DVP_ST_WEAK_UPDATE>(e, iter, colour); break; }
		const int S = e.NI - 1;
		const size_t n = (size_t)std::max(e.d.weak_count, 1);
		WeakRec poison;
		std::memset(&poison, 0xA5, sizeof(poison));   // (nothing may read a field no launch of THIS update wrote)
		e.weak_rec.assign(n, poison);
		e.weak_ctab.assign(n * kTaps * kTaps, mk2(-7.0f, -7.0f));
		e.weak_ev.assign(n * 8 * S, -7.0f);
		refresh(e);
		const LaunchGeom g = make_geom(e.W, e.H, true);
		// the WEAK pixels of the launch in block order (the engine: the WEAK list of the colour), taken `group` at a time by the
		// evaluation launches
		std::vector<int> list;
		for (int b = 0; b < g.grid(); ++b)
			for (int wave = 0; wave < 4; ++wave)
				for (int lane = 0; lane < 64; ++lane) {
					int px, py;
					if (block_to_pixel(b, lane, wave, g.tiles_x, g.tiles, g.rows, 1, colour, e.W, e.H, &px, &py) && e.d.weak_info[px + py * e.W] == DVP_WEAK) list.push_back(px + py * e.W);
				}
		int group[4] = { 1, 4, 4, 2 };
		if (const char* gs = getenv("DVP_WEAK_GROUPS")) {
			int q[4];
			if (sscanf(gs, "%d,%d,%d,%d", &q[0], &q[1], &q[2], &q[3]) == 4)
				for (int i = 0; i < 4; ++i) group[i] = q[i] < 1 ? 1 : (q[i] > kGrp ? kGrp : q[i]);
		}
		const bool ex = e.d.sampler != 0, u8 = e.d.images8 != nullptr;
		const long long n_px = (long long)list.size();
		unsigned long long total = 0;
		auto eval = [&](int mode) {
			const int G = std::min(group[mode], mode == 0 ? kGrpWide : kGrp);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
			for (long long t0 = 0; t0 < n_px; t0 += G) {
				unsigned long long k = 0;
				unsigned long long* kp = e.count ? &k : nullptr;
#define EMU_GROUP(M, GRP) { WeakGroupSharedT<GRP> sh; for (int i = 0; i < GRP; ++i) sh.center[i] = (i < G && t0 + i < n_px) ? list[(size_t)(t0 + i)] : -1; \
	if (ex) { if (u8) weak_group_eval<1, 1, M, GRP>(e.d, G, kp, sh); else weak_group_eval<1, 0, M, GRP>(e.d, G, kp, sh); } else { if (u8) weak_group_eval<0, 1, M, GRP>(e.d, G, kp, sh); else weak_group_eval<0, 0, M, GRP>(e.d, G, kp, sh); } }
				switch (mode) { case 0: EMU_GROUP(0, kGrpWide) break; case 1: EMU_GROUP(1, kGrp) break; case 2: EMU_GROUP(2, kGrp) break; default: EMU_GROUP(3, kGrp) break; }
#undef EMU_GROUP
				total += k;
			}
		};
		auto each = [&](int part) {
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
			for (long long t = 0; t < n_px; ++t) {
				const int c0 = list[(size_t)t], py = c0 / e.W, px = c0 - py * e.W;
				unsigned long long k = 0;
				f2 tab_mem[kTaps * kTaps];
				const PatchTab tab{tab_mem, 1};
				switch (part) {
				case 1: weak_d1_px(e.d, px, py, iter); break;
				case 2: weak_d2_px(e.d, px, py, iter); break;
				case 3: weak_d3_px(e.d, px, py); break;
				default: if (ex) weak_final_cost_px<1>(e.d, px, py, tab, e.count ? &k : nullptr); else weak_final_cost_px<0>(e.d, px, py, tab, e.count ? &k : nullptr); break;
				}
				total += k;
			}
		};
		eval(0); each(1); eval(1); each(2); eval(2); eval(3); each(3); each(4);
		e.evals += total;
		break;
	}
	case DVP_ST_GET_DEPTH_NORMAL: launch<DVP_ST_GET_DEPTH_NORMAL>(e, iter, colour); break;
	case DVP_ST_FILTER_STRONG: launch<DVP_ST_FILTER_STRONG>(e, iter, colour); break;
	case DVP_ST_DEPTH_TO_WEAK: launch<DVP_ST_DEPTH_TO_WEAK>(e, iter, colour); break;
	case DVP_ST_LOCAL_REFINE: launch<DVP_ST_LOCAL_REFINE>(e, iter, colour); break;
	case kStageSweeps: {
		// the engine issues the fused launch site as view-compacted passes unless DVP_SWEEP_SPLIT=0; the emulation follows the switch
		const char* sp = getenv("DVP_SWEEP_SPLIT");
		if ((sp && atoi(sp) == 0) || !(e.d.params.geom_consistency || (sp && atoi(sp) == 2))) { launch<kStageSweeps>(e, iter, colour); break; }   // (the engine's rule: the passes where the geometric term is on, DVP_SWEEP_SPLIT=2 everywhere)
		const size_t L = (size_t)e.W * e.H;
		const int S = e.NI - 1;
		e.sweep_rec.assign(2 * L, mk4(0, 0, 0, 0));
		e.sweep_cost.assign(sweep_cost_floats(L, S), -7.0f);   // (a value no decision may ever read)
		e.sweep_pc.assign(61 * L, -7.0f);
		refresh(e);
		const long long n = (long long)L;
		auto each_pixel = [&](auto&& f) {
#pragma omp parallel for schedule(dynamic, 256)
			for (long long c = 0; c < n; ++c) f((int)(c % e.W), (int)(c / e.W));
		};
		each_pixel([&](int px, int py) { sweep_prepare_px(e.d, px, py); });
		for (int stage = 0; stage < 2; ++stage) {
			if (stage == 1 && sweep_window(e.d.params) >= 30) break;
			unsigned long long total = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : total)
			for (long long c = 0; c < n; ++c)
				for (int v = 0; v < S; ++v) {
					if (!sweep_go(e.d, (int)c, v, stage)) continue;
					unsigned long long k = 0;
					f2 tab_mem[kTaps * kTaps];
					const PatchTab tab{tab_mem, 1};
					if (e.d.sampler) sweep_eval_px<1>(e.d, (int)(c % e.W), (int)(c / e.W), v, stage, tab, e.count ? &k : nullptr);
					else sweep_eval_px<0>(e.d, (int)(c % e.W), (int)(c / e.W), v, stage, tab, e.count ? &k : nullptr);
					total += k;
				}
			e.evals += total;
			if (stage == 0) each_pixel([&](int px, int py) { sweep_decide1_px(e.d, px, py); });
		}
		each_pixel([&](int px, int py) { sweep_decide2_px(e.d, px, py); });
		launch<kStageSweeps>(e, kSweepBorderOnly, colour);
		break;
	}
	default: return -1;
	}
	return 0;
}

int emu_run_patchmatch(void* c) {
	Emu& e = *(Emu*)c;
	emu_run_stage(c, DVP_ST_GEN_EDGE_INFORM, 0, 0);
	emu_run_stage(c, DVP_ST_FIND_NEAREST_STRONG, 0, 0);
	emu_run_stage(c, DVP_ST_GEN_NEIGHBOURS, 0, 0);
	emu_run_stage(c, DVP_ST_NEIGHBOUR_UPDATE, 0, 0);
	emu_run_stage(c, DVP_ST_RANDOM_INIT, 0, 0);
	for (int i = 0; i < e.d.params.max_iterations; ++i) {
		emu_run_stage(c, DVP_ST_STRONG_UPDATE, i, 0);
		emu_run_stage(c, DVP_ST_STRONG_UPDATE, i, 1);
		emu_run_stage(c, DVP_ST_RANSAC_FIT, i, 0);
		emu_run_stage(c, DVP_ST_WEAK_UPDATE, i, 0);
		emu_run_stage(c, DVP_ST_WEAK_UPDATE, i, 1);
	}
	emu_run_stage(c, DVP_ST_GET_DEPTH_NORMAL, 0, 0);
	emu_run_stage(c, DVP_ST_FILTER_STRONG, 0, 0);
	emu_run_stage(c, DVP_ST_FILTER_STRONG, 0, 1);
	emu_run_stage(c, kStageSweeps, 0, 0);   // DepthToWeak + LocalRefine in one launch, as dvp_run_patchmatch does
	return 0;
}

// the line test of the weak path, both forms (dvp_weak.hpp): which = 0 the closed-form block skipper the kernels call, 1 the
// definition that makes every step; the edge map is the context's (packed here)
int emu_line_test(void* c, int ax, int ay, int bx, int by, int which) {
	Emu& e = *(Emu*)c;
	return (which ? bresenham_hits_edge_steps(e.d, ax, ay, bx, by) : bresenham_hits_edge(e.d, ax, ay, bx, by)) ? 1 : 0;
}
void emu_pack_edges(void* c) { pack_edge(*(Emu*)c); }
float emu_expf(float x) { return dvp_expf(x); }
void emu_eval_cost_vectors(void* c, const int* px, const float* planes, int n, float* out) {
	Emu& e = *(Emu*)c;
	const int S = e.NI - 1;
#pragma omp parallel for schedule(dynamic, 64)
	for (int i = 0; i < n; ++i) {
		const int x = px[2 * i], y = px[2 * i + 1];
		PatchCtx pc;
		f2 tab_mem[kTaps * kTaps];
		int radius, inc;
		patch_geometry(e.d, x + y * e.W, &radius, &inc);
		build_patch_ctx(e.d, x, y, radius, inc, 0, PatchTab{tab_mem, 1}, &pc);
		for (int v = 0; v < S; ++v)
			out[(size_t)i * S + v] = (e.d.sampler ? ncc_old<1> : ncc_old<0>)(e.d, pc, x, y, v + 1, mk4(planes[4 * i], planes[4 * i + 1], planes[4 * i + 2], planes[4 * i + 3]));
	}
}

}  // extern "C"
