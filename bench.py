#!/usr/bin/env python3
"""bench.py — Mpixels/s/PatchMatch-iteration of the MI355X PatchMatch engine.

A "step" = one APD::RunPatchMatch (reference: /root/reference/APD.cu:4406-4532) over one reference
view of synthetic input.  Default workload (N=1) = BASELINE.json configs[2], the configuration the
metric is quoted on ("ETH3D full-res"): 6208x4128, 9 source views, a REFINE_ITER pass with
geometric consistency, use_APD (WEAK pixels) and the edge/label/radius priors on — all twelve launch
sites live.  Its inputs (planes, selected views, WEAK map, radius map) come from an UNTIMED FIRST_INIT
pass of the same view on the same engine, handed over the way the reference's driver does between
passes (main.cpp:298-376 -> APD.cpp:1169-1195,1428-1456); the source depth maps are the scene's
rendered depths.  `--config cfg2` selects the half-res FIRST_INIT configuration (round-1 bench line).

With N>1 (torchrun, one rank per GPU): rank 0 renders the scene on its GPU, images / depth maps /
surface maps are broadcast once over RCCL (xGMI); rank r then takes view (r mod NI) of the scene as
ITS reference view (the other NI-1 as sources) — different content per rank, no data-path collective,
weak scaling.  value = W*H*iters*steps*N / max-over-ranks wall time of the timed steps (whole
RunPatchMatch incl. the device-side state restore; inputs resident in HBM).

Prints ONE JSON line (rank 0).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NCC_BYTES = 724            # algorithmic bytes of one bilateral-NCC evaluation
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
# fp32 VALU issue peak of MI355X per /opt/skills/guides/MI355X_MICROARCH.md (v_fma_f32 wave64 = 2 cycles per SIMD):
# 256 CU x 4 SIMD x 2.4 GHz / 2 = 1228.8 G wave64 instructions/s.  `roofline.frac` of a VALU-bound kernel is quoted
# against THIS number.  The rate MEASURED on the part with tools/valu_peak.hip (independent v_fma_f32 chains,
# profiles/r02_valu_peak.txt) is lower and depends on occupancy: 977.5 G/s with 8 waves per SIMD, 761.6 with the 2 waves
# per SIMD the NCC kernels hold (256 VGPRs + a 72 KB patch table per workgroup); both are reported as extra fields.
VALU_PEAK_GINST = 1228.8
VALU_MEASURED_PEAK_GINST = 977.5
VALU_CEILING_BY_WAVES = {1: 501.6, 2: 761.6, 3: 852.3, 4: 896.1, 6: 943.3, 8: 977.5}
GATHER_ROOF_GLINES = 51.4   # tools/gather_peak.hip on MI355X (profiles/r02_gather_peak.txt): 128-byte line requests per second of 16-byte gathers = 6.6 TB/s
WAVES_PER_SIMD = {"strong_update": 2, "depth_to_weak": 2, "local_refine": 2, "random_init": 2, "weak_update": 4}
PMC_TABLE = os.path.join(ROOT, "profiles", "pmc_r06.json")
EVALUATOR_PEAK_GEVALS = 32.7   # the 36-tap evaluator alone at 2 waves per SIMD (tools/pv_probe.py, profiles/r03_pv_probe.txt): what a launch site of NCC evaluations can reach at best


def csrc_sha256():
    """Hash of the kernel sources in the tree (tools/csrc_hash.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import csrc_hash
    return csrc_hash.csrc_sha256()


_PMC_CACHE = {}
RIG = "rotated"
SRC_DEPTHS = "estimated"
LOADED_BUILD_ID = None    # dvp_build_id() of the library the timed context runs on (set in main)


def pmc_table():
    """(table or None, note).  The counters are only used when the table was collected on exactly the kernel sources the
    LOADED library was built from (the library embeds their hash); anything else — no table, another tree, a variant
    build — gives None and the bench line says why."""
    if "t" in _PMC_CACHE:
        return _PMC_CACHE["t"]
    res = (None, "no PMC table (%s)" % os.path.relpath(PMC_TABLE, ROOT))
    try:
        t = json.load(open(PMC_TABLE))
        if LOADED_BUILD_ID is None:
            res = (None, "PMC table not consulted: library build id unknown")
        elif t.get("csrc_sha256") != LOADED_BUILD_ID:
            res = (None, "PMC table refused: collected on csrc sha256 %s, the loaded library was built from %s" % (str(t.get("csrc_sha256"))[:12], LOADED_BUILD_ID[:24]))
        else:
            res = (t, "csrc sha256 %s" % LOADED_BUILD_ID[:12])
    except Exception as e:   # missing / unreadable table
        res = (None, "no usable PMC table: %s" % e)
    _PMC_CACHE["t"] = res
    return res

# launch site -> kernel name in a rocprofv3 trace (list launches for the weak path; the narrow
# strong-update instantiation when S <= 8)
IMAGE_FORMAT = 0   # dvp_image_format of the timed context (1: the weak update reads byte planes)


def decide_kernel(S):
    return "dvp_strong_decide_v%d" % next(m for m in (4, 6, 8, 10, 12, 16) if S <= m)


GEOM = True   # the timed pass has the geometric term on (set in main)


def sweep_split():
    """DepthToWeak + LocalRefine run as view-compacted passes where the geometric term is on (dvp_engine.hip: launch_stage)"""
    e = os.environ.get("DVP_SWEEP_SPLIT", "1")
    return e != "0" and (GEOM or e == "2")


WEAK_COUNT = 1 << 30   # WEAK pixels of the timed context (set in measure)


def weak_one_launch_site():
    """dvp_run_patchmatch issues both colours of a weak update as ONE launch site (colour 2) under this condition (dvp_engine.hip)"""
    return os.environ.get("DVP_WEAK_PHASED", "1") != "0" and os.environ.get("DVP_WEAK_ANCHOR_TAB", "1") != "0" and "DVP_WEAK_SPLIT_COLOURS" not in os.environ


def weak_phased():
    """the weak update runs as evaluation launches (a wave per group of WEAK pixels) + decision launches (a lane per WEAK pixel)
    where the anchor table is on and the launch holds at least DVP_WEAK_PHASED_MIN (8192) WEAK pixels; below that — and when the
    switch is off — the one-wave kernel takes them (dvp_engine.hip: launch_stage)"""
    tmin = int(os.environ.get("DVP_WEAK_PHASED_MIN", "8192"))
    return os.environ.get("DVP_WEAK_PHASED", "1") != "0" and os.environ.get("DVP_WEAK_ANCHOR_TAB", "1") != "0" and (tmin <= 0 or WEAK_COUNT >= tmin)


def extra_kernels(stage, S):
    """the other kernels a launch site's timer covers: [(kernel, launches per launch of the site)]; their counters are added to
    the first one's"""
    if stage == "gen_neighbours":
        return [("dvp_gen_neighbours_fit", 1)]       # (dvp_gen_candidates_views runs on the side stream, outside its launch site's timer)
    if stage == "strong_update" and S <= 16 and os.environ.get("DVP_STRONG_SPLIT", "1") != "0":
        return [(decide_kernel(S), 1), ("dvp_strong_refine_lanes" if os.environ.get("DVP_REFINE_LANES", "1") != "0" else "dvp_strong_refine", 1)]
    if stage == "weak_update" and weak_phased():     # the weak update as eight launches (dvp_weak_phased.hpp)
        u8 = "_u8" if IMAGE_FORMAT else ""
        return [("dvp_weak_select_views", 1), ("dvp_weak_eval_planes" + u8, 1), ("dvp_weak_make_hypotheses", 1), ("dvp_weak_eval_first_view" + u8, 1),
                ("dvp_weak_eval_survivors" + u8, 1), ("dvp_weak_adopt", 1), ("dvp_weak_final_cost", 1)]
    if stage == "depth_to_weak" and sweep_split():   # DepthToWeak + LocalRefine as view-compacted passes
        return [("dvp_sweep_prepare", 1), ("dvp_sweep_decide1", 1), ("dvp_sweep_decide2", 1), ("dvp_sweep_border", 1)] + \
               []
    return []


WEAK_PEAK_RADIUS = 4   # of the timed pass (set in measure)


def primary_launches(stage):
    """launches of the site's first kernel per launch of the site: dvp_sweep_eval runs its second stage only when the central
    window leaves slots over (dvp_engine.hip: sweep_window(params) < 30, i.e. weak_peak_radius + 1 < 30)"""
    if stage == "depth_to_weak" and sweep_split():
        return 2 if min(max(WEAK_PEAK_RADIUS + 1, 5), 30) < 30 else 1
    return 1


def kernel_name(stage, S):
    split = S <= 16 and os.environ.get("DVP_STRONG_SPLIT", "1") != "0"
    return {"strong_update": ("dvp_strong_eval_items" if os.environ.get("DVP_EVAL_ITEMS", "1") != "0" else "dvp_strong_eval") if split else ("dvp_strong_update_v8" if S <= 8 else ("dvp_strong_update_v16" if S <= 16 else "dvp_strong_update")),
            "weak_update": ("dvp_weak_eval_candidates" + ("_u8" if IMAGE_FORMAT else "")) if weak_phased() else
                           (("dvp_weak_update_wave_u8" if IMAGE_FORMAT else "dvp_weak_update_wave") + ("" if os.environ.get("DVP_WEAK_ANCHOR_TAB", "1") != "0" else "_notab")),
            "depth_to_weak": "dvp_sweep_eval" if sweep_split() else "dvp_depth_to_weak_refine",   # dvp_run_patchmatch: DepthToWeak + LocalRefine as one launch site
            "gen_neighbours": "dvp_gen_neighbours_search" if os.environ.get("DVP_GN_WAVE", "0") not in ("", "0") else "dvp_gen_neighbours_list",
            "ransac_fit": "dvp_ransac_fit_plane_wave" if os.environ.get("DVP_RANSAC_WAVE", "0") not in ("", "0") else "dvp_ransac_fit_plane_list", "find_nearest_strong": "dvp_find_nearest_strong_list",
            "neighbour_update": "dvp_neighbour_update_list"}.get(stage, "dvp_" + stage)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--src", type=int, default=0)
    ap.add_argument("--iters", type=int, default=0)
    ap.add_argument("--weak-frac", type=float, default=None, help="share of 32x32 tiles handed over as WEAK (refine configs); default 0.05, cfg5: 0.10 (SURVEY 8d: >= 10 %% WEAK)")
    ap.add_argument("--weak-layout", default="tiles", choices=["tiles", "regions"], help="shape of the pixels handed over as WEAK: 32x32 tiles (default) or a few large connected regions (workloads.weak_regions)")
    ap.add_argument("--rig", default="rotated", choices=["rotated", "axis"], help="camera rig of the synthetic scene: per-view rotations and intrinsics (default) or the round-1/2 rig (R = I, one K)")
    ap.add_argument("--src-depths", default="estimated", choices=["estimated", "gt"], help="depth maps the geometric term reads: 'estimated' = the rendered depths with 0.3 %% relative noise, 2 %% of 16x16 blocks and 1 %% of single pixels missing (depth 0), as maps estimated by a previous pass are; 'gt' = the rendered depths (rounds 1-4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE", help="override a PatchMatchParams field of the timed pass (ablations, e.g. --param use_limit=0); the line's workload string says so")
    ap.add_argument("--no-per-iteration", action="store_true", help="skip the extra untimed pass that times the iteration loop iteration by iteration (PMC / trace runs: keeps the dispatch list to the timed steps)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the cfg5 / cfg2 lines that the default workload appends")
    ap.add_argument("--dry-launch", action="store_true", help="launcher check without a GPU: start the ranks, rendezvous over gloo, print one line with each rank's environment")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself (0: pick a free one)")
    ap.add_argument("--cpu-size", type=str, default="auto", help="WxH of the CPU-baseline view (auto: scaled to the core count)")
    return ap.parse_args()


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    omp = os.environ.get("OMP_NUM_THREADS")
    return min(n, int(omp)) if omp else n


def cpu_baseline(pkg, args, cfg, S, iters, device=0):
    """The oracle ("port") timed on this host's cores on a bounded sample of the same workload: same
    scene generator, same S / iterations / parameters / pass structure on a small view; the engine is
    then run on that very sample and compared with it (oracle = checker, outside any timed region)."""
    from oracle import oracle as O
    synth, wl, capi = pkg.synth, pkg.workloads, pkg.get_capi()
    ncores = host_cores()
    if args.cpu_size == "auto":
        # ~0.0006 Mpx*iter/s/core measured (r01): ~1500 px per core keeps the timed pass at 10-30 s
        px = max(192 * 144, 1500 * ncores)
        h = int(round((px * 3 / 4) ** 0.5 / 8)) * 8
        w = h * 4 // 3
    else:
        w, h = [int(v) for v in args.cpu_size.split("x")]
    sc = synth.make_scene(w, h, S, rig=args.rig)
    L = w * h
    p1 = wl.first_init_params(S, iters)
    first = dict(planes=np.zeros((L, 4), np.float32), edge=sc["edge"], label=sc["label"], radius=np.full(L, 5, np.int32))
    o = O.from_scene(sc, p1)
    g = capi.from_scene(sc, p1, device=device)
    o.upload_state(**first)
    g.upload_state(**first)
    t0 = time.time()
    o.run_patchmatch()
    t_first = time.time() - t0
    g.run_patchmatch()
    timed, what = t_first, "FIRST_INIT pass"
    if cfg["refine"]:
        p2 = wl.refine_iter_params(S, iters)
        st = wl.hand_over(o.get("planes"), o.get("selected_views"), o.get("weak_info"), o.get("radius"), p1, w, h,
                          extra_weak=wl.weak_mask(args.weak_layout, w, h, args.weak_frac, sc["flat"]))
        deps = np.array(sc["depth_gt"], np.float32, copy=True)
        if args.src_depths == "estimated":   # the same kind of source depth maps as the timed workload: noise + holes
            r = np.random.default_rng(20240904)
            deps *= (1.0 + 0.003 * r.standard_normal(deps.shape)).astype(np.float32)
            blocks = r.random((deps.shape[0], (h + 15) // 16, (w + 15) // 16)) < 0.02
            holes = np.repeat(np.repeat(blocks, 16, 1), 16, 2)[:, :h, :w] | (r.random(deps.shape) < 0.01)
            deps[holes] = 0.0
        for e in (o, g):
            e.set_params(p2)
            e.set_depths(deps)
            e.upload_state(planes=st[0], views=st[1], weak=st[2], radius=st[3])
        t0 = time.time()
        o.run_patchmatch()
        timed, what = time.time() - t0, "REFINE_ITER pass (geom against %s source depth maps, %.1f %% WEAK) after an untimed FIRST_INIT pass" % (args.src_depths, 100.0 * o.weak_count() / L)
        g.run_patchmatch()
    res = {"value": round(L * iters / timed / 1e6, 5), "unit": "Mpx/s/iter", "cores": ncores, "kind": "port",
           "sample": "this repo's CPU restatement of the path (oracle/, g++ -O3, OpenMP over row blocks) — NOT the reference rebuilt for CPU, which cannot be built here (it needs nvcc, OpenCV, Boost): %dx%d view, S=%d, %d iterations, whole RunPatchMatch of the %s, %.1f s" % (w, h, S, iters, what, timed)}
    a, b = o.get("planes"), g.get("planes")
    diff = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    res["gpu_vs_cpu_plane_words_differing"] = int(diff.sum())
    res["gpu_vs_cpu_states_differing"] = int((o.get("weak_info") != g.get("weak_info")).sum() + (o.get("selected_views") != g.get("selected_views")).sum())
    g.close()
    return res


def pmc_lookup(kernel, W, H, S):
    """PMC counters of `kernel` at exactly this problem size (profiles/pmc_r06.json, written by
    tools/pmc_table.py from separate rocprofv3 --pmc passes of this bench ON THE SAME KERNEL SOURCES) or None."""
    t, _ = pmc_table()
    if RIG != "rotated" or SRC_DEPTHS != "estimated":     # the table is collected on the default workload
        return None
    return t["kernels"].get("%s|%dx%d|S%d" % (kernel, W, H, S)) if t else None


def roofline_of(stage, S, W, H, avg_ms, evals_per_launch):
    """Roofline entry of one launch site; a reader can recompute every fraction from profiles/ alone:
      frac (bound "valu")  = SQ_INSTS_VALU per launch (profiles/pmc_r06.json) / live launch time / 1228.8 G/s
                             (guide peak: 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 v_fma_f32);
      valu_frac_of_measured_issue_peak = same rate / 977.5 G/s (tools/valu_peak.hip, 8 waves per SIMD);
      valu_frac_of_occupancy_ceiling   = same rate / the measured ceiling at the kernel's own waves per SIMD;
      hbm.frac = physical traffic (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md) / time / 8 TB/s;
      work_rate = SURVEY 8(d)'s cache-oblivious figure: NCC evaluations x 724 B / time.  It is a WORK RATE, not a
                  roofline: 148 of the 724 B never leave LDS (hoisted reference side) and L1/L2/MALL serve most of
                  the rest, so it exceeds the HBM peak (1.5-2x) — the north_star's ">= 70 % of HBM roofline on the
                  NCC kernel" is met trivially by that definition and says nothing; the VALU fraction does.
    `bound` = whichever of valu / hbm has the larger fraction.  Counters are only used when the PMC table was
    collected on the kernel sources in the tree (csrc sha256), else traffic is null."""
    k = kernel_name(stage, S)
    sec = avg_ms * 1e-3
    alg = evals_per_launch * NCC_BYTES / sec / 1e9 if (sec > 0 and evals_per_launch) else None
    pmc = pmc_lookup(k, W, H, S)
    SUMMED = ("hbm_bytes_per_launch", "SQ_INSTS_VALU", "TCC_MISS", "TCC_HIT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "TCP_TOTAL_CACHE_ACCESSES", "TCP_TCC_READ_REQ")
    if pmc and primary_launches(stage) != 1:   # the table holds per-launch averages: a site that launches a kernel twice has twice of each
        pmc = dict(pmc)
        for c in SUMMED:
            if c in pmc:
                pmc[c] = pmc[c] * primary_launches(stage)
    for second, mult in (extra_kernels(stage, S) if pmc else []):   # a launch site with several kernels is timed as one: its counters are their sum
        p2 = pmc_lookup(second, W, H, S)
        if p2:
            pmc = dict(pmc)
            for c in SUMMED:
                if c in pmc and c in p2:
                    pmc[c] = pmc[c] + mult * p2[c]
            k = k + " + " + second
    if pmc:
        # lane utilisation of the site = active-lane cycles / VALU-busy cycles x 64, summed over its kernels
        parts = [(pmc_lookup(kernel_name(stage, S), W, H, S), primary_launches(stage))] + [(pmc_lookup(e, W, H, S), m) for e, m in extra_kernels(stage, S)]
        tc = sum(m * q["SQ_THREAD_CYCLES_VALU"] for q, m in parts if q and q.get("lane_utilisation"))
        act = sum(m * q["SQ_THREAD_CYCLES_VALU"] / q["lane_utilisation"] for q, m in parts if q and q.get("lane_utilisation"))
        pmc["lane_utilisation"] = round(tc / act, 4) if act else None
        if pmc.get("TCC_HIT") is not None and pmc.get("TCC_MISS"):
            pmc["l2_hit_rate"] = round(pmc["TCC_HIT"] / (pmc["TCC_HIT"] + pmc["TCC_MISS"]), 4)
        if pmc.get("TCP_TOTAL_CACHE_ACCESSES") and pmc.get("TCP_TCC_READ_REQ") is not None:
            pmc["l1_hit_rate"] = round(1.0 - min(1.0, pmc["TCP_TCC_READ_REQ"] / pmc["TCP_TOTAL_CACHE_ACCESSES"]), 4)
        if pmc.get("SQ_WAVE_CYCLES") and pmc.get("SQ_WAIT_ANY") is not None:
            pmc["wait_any_frac"] = round(pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"], 4)
    gev = evals_per_launch / sec / 1e9 if (sec > 0 and evals_per_launch) else None
    r = {"kernel": k, "avg_launch_ms": round(avg_ms, 3), "evals_per_launch": int(evals_per_launch or 0),
         # what the lanes did: evaluations per second against the evaluator running alone (36-tap evaluations; the weak
         # update's are 135-tap evaluations of another evaluator, its figure is not comparable and left out)
         "Gevals_per_s": round(gev, 2) if gev else None,
         "useful_eval_rate_frac": round(gev / EVALUATOR_PEAK_GEVALS, 3) if (gev and stage != "weak_update") else None,
         "work_rate": {"what": "SURVEY 8(d) algorithmic bytes: NCC evaluations x 724 B / launch time — a work rate, NOT a bound (caches/LDS serve it)",
                       "bytes_per_eval": NCC_BYTES, "gbs": round(alg, 1) if alg else None,
                       "over_hbm_peak": round(alg / HBM_PEAK_GBS, 4) if alg else None},
         # SURVEY 8(d)'s own figure under its own name, next to `frac`: algorithmic bytes / time / 8 TB/s
         "survey_8d_frac_of_hbm_peak": round(alg / HBM_PEAK_GBS, 4) if alg else None}
    if pmc and sec > 0:
        traffic = pmc.get("hbm_bytes_per_launch")
        insts = pmc.get("SQ_INSTS_VALU")
        hbm_gbs = traffic / sec / 1e9 if traffic else None
        ginst = insts / sec / 1e9 if insts else None
        fh = hbm_gbs / HBM_PEAK_GBS if hbm_gbs else 0.0
        fv = ginst / VALU_PEAK_GINST if ginst else 0.0
        if fv >= fh:
            r.update(bound="valu", achieved=round(ginst, 1), peak=VALU_PEAK_GINST, unit="G wave64 VALU instr/s", frac=round(fv, 4))
        else:
            r.update(bound="hbm", achieved=round(hbm_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(fh, 4))
        wps = WAVES_PER_SIMD.get(stage)
        r.update(traffic=traffic,
                 hbm={"achieved": round(hbm_gbs, 1) if hbm_gbs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fh, 4)},
                 physical_hbm_frac=round(fh, 4),
                 valu={"achieved": round(ginst, 1) if ginst else None, "peak": VALU_PEAK_GINST, "unit": "G wave64 VALU instr/s", "frac": round(fv, 4)},
                 valu_frac=round(fv, 4),
                 valu_frac_of_measured_issue_peak=round(ginst / VALU_MEASURED_PEAK_GINST, 4) if ginst else None,
                 waves_per_simd=wps,
                 valu_frac_of_occupancy_ceiling=round(ginst / VALU_CEILING_BY_WAVES.get(wps, VALU_MEASURED_PEAK_GINST), 4) if (ginst and wps in VALU_CEILING_BY_WAVES) else None,
                 valu_instr_per_launch=insts,
                 valu_instr_per_wave_eval=round(insts * 64.0 / evals_per_launch, 1) if (insts and evals_per_launch) else None,
                 pmc_source="profiles/%s[%s|%dx%d|S%d], %s" % (os.path.basename(PMC_TABLE), k, W, H, S, pmc_table()[1]))
        for c in ("l2_hit_rate", "l1_hit_rate", "wait_any_frac", "lane_utilisation"):
            if c in pmc:
                r[c] = pmc[c]
        if traffic and evals_per_launch:
            r["hbm_bytes_per_eval"] = round(traffic / evals_per_launch, 1)
        if pmc.get("TCC_MISS"):   # L2 line misses per second against the measured gather roof
            r["l2_miss_glines_s"] = round(pmc["TCC_MISS"] / sec / 1e9, 2)
            r["gather_roof_frac"] = round(pmc["TCC_MISS"] / sec / 1e9 / GATHER_ROOF_GLINES, 4)
    else:
        r.update(bound="valu", achieved=None, peak=VALU_PEAK_GINST, unit="G wave64 VALU instr/s", frac=None, traffic=None,
                 note="no counters for this (kernel, size): %s — launch time and work rate only" % pmc_table()[1])
    return r


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this very command line, one per GPU, under
    torch.distributed.run (the same form the driver uses when it launches the ranks itself) and pass rank 0's JSON
    line through on stdout.  The reference has nothing to mirror here: /root/reference/main.cpp:430-434 picks ONE
    device per process."""
    import subprocess
    if not args.dry_launch:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s) (torch.cuda.device_count()); refusing to measure fewer ranks than asked for" % (args.gpus, have))
    port = args.master_port or free_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_launch(args, json_fd, rank, local_rank, world):
    """Launcher check that needs no GPU: every rank joins a gloo group and reports the environment it was started
    with; rank 0 prints the line (tests/test_sharding.py::test_bench_gpus_flag_starts_the_ranks)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    mine = {"rank": rank, "local_rank": local_rank, "world_size": world, "pid": os.getpid(),
            "device": "cuda:%d" % local_rank, "master": "%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])}
    seen = [mine]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, mine)
    if rank == 0:
        os.write(json_fd, (json.dumps({"metric": "Mpixels/sec/PatchMatch-iteration", "value": None, "dry_launch": True,
                                       "n_gpus": world, "gpus_flag": args.gpus, "ranks": seen}) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def per_iteration_ms(ctx, iters, view_index, weak):
    """SURVEY 7 (hard part 4): early iterations gather worse than converged ones — the launch sites of the iteration loop
    (APD.cu:4478-4492) timed iteration by iteration in one extra, untimed pass issued stage by stage (dvp_run_stage) with the
    launch structure dvp_run_patchmatch uses: both colours of a weak update as one launch site (colour 2) where it does so."""
    ctx.set_seed(1234 + view_index)
    ctx.set_profiling(False)
    ctx.restore_state()
    for st in ("gen_edge_inform", "find_nearest_strong", "gen_neighbours", "neighbour_update", "random_init"):
        ctx.run_stage(st, 0, 0)
    ctx.synchronize()
    ctx.timings(reset=True)
    res = []
    for it in range(iters):
        ctx.run_stage("strong_update", it, 0)
        ctx.run_stage("strong_update", it, 1)
        if weak:
            ctx.run_stage("ransac_fit", it, 0)
            if weak_one_launch_site():
                ctx.run_stage("weak_update", it, 2)
            else:
                ctx.run_stage("weak_update", it, 0)
                ctx.run_stage("weak_update", it, 1)
        ctx.synchronize()
        t = ctx.timings(reset=True)
        res.append({k: round(v, 3) for k, v in t["stage_ms"].items() if v > 0})
    return res


def measure(env, args, cfg_name, W, H, S, iters, weak_frac_arg, steps, warmup, primary):
    """one workload on this rank's GPU: scene, context, (untimed FIRST_INIT + hand-over), warm-up, timed steps -> the line's dict (rank 0)"""
    torch, dist, dev = env["torch"], env["dist"], env["dev"]
    rank, local_rank, world, use_dist, pkg = env["rank"], env["local_rank"], env["world"], env["use_dist"], env["pkg"]
    synth, wl, capi = pkg.synth, pkg.workloads, pkg.get_capi()
    cfg = dict(wl.CONFIGS[cfg_name])
    NI, L = S + 1, W * H
    _PMC_CACHE.pop("t", None)

    # ---- inputs: rank 0 renders the scene on its GPU; every rank gets it over RCCL (xGMI) ----------
    t_setup = time.time()
    imgs = torch.empty((NI, H, W), dtype=torch.float32, device=dev)
    deps = torch.empty((NI, H, W), dtype=torch.float32, device=dev)
    sids = torch.empty((NI, H, W), dtype=torch.uint8, device=dev)
    flats = torch.empty((NI, H, W), dtype=torch.bool, device=dev)
    cams_t = torch.empty(NI * 112, dtype=torch.uint8, device=dev)
    if rank == 0:
        sc = synth.make_scene_torch(W, H, S, dev, rig=args.rig)
        imgs.copy_(sc["images"])
        deps.copy_(sc["depth_gt"])
        if args.src_depths == "estimated":
            # what a previous pass hands over is not the truth: noise, holes (filtered pixels are stored as depth 0, which the
            # geometric term answers with its maximum, APD.cu:1238) — VERDICT r03 weak #9
            gen = torch.Generator(device=dev)
            gen.manual_seed(20240904)
            deps.mul_(1.0 + 0.003 * torch.randn(deps.shape, generator=gen, device=dev))
            blocks = torch.rand((NI, (H + 15) // 16, (W + 15) // 16), generator=gen, device=dev) < 0.02
            holes = blocks.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :H, :W] | (torch.rand(deps.shape, generator=gen, device=dev) < 0.01)
            deps.masked_fill_(holes, 0.0)
            del blocks, holes
        for i in range(NI):
            sids[i], flats[i] = sc["sids"][i], sc["flats"][i]
        cams_t.copy_(torch.from_numpy(np.frombuffer(sc["cameras"].tobytes(), np.uint8).copy()))
        del sc
    if use_dist:
        for t in (imgs, deps, sids, cams_t):
            dist.broadcast(t, 0)
        fl8 = flats.to(torch.uint8)
        dist.broadcast(fl8, 0)
        flats = fl8.bool()
    torch.cuda.synchronize()
    cams_all = np.frombuffer(cams_t.cpu().numpy().tobytes(), dtype=synth.CAMERA_DTYPE).copy()

    # ---- this rank's problem: view `ref` is the reference image, the others its sources -------------
    ref = rank % NI
    order = [ref] + [i for i in range(NI) if i != ref]
    cams = cams_all[order].copy()
    edge_t, label_t = synth.view_priors_torch(sids[ref], flats[ref])
    edge, label = edge_t.cpu().numpy(), label_t.cpu().numpy()
    flat = flats[ref].cpu().numpy()
    del sids, flats, edge_t, label_t

    ctx = capi.Context(W, H, NI, device=local_rank)
    global LOADED_BUILD_ID, RIG, GEOM, SRC_DEPTHS
    RIG = args.rig
    SRC_DEPTHS = args.src_depths
    GEOM = bool(cfg["refine"])
    LOADED_BUILD_ID = ctx.L.dvp_build_id().decode()
    ctx.set_images_device([imgs[i].data_ptr() for i in order], W)
    global IMAGE_FORMAT
    IMAGE_FORMAT = ctx.image_format()   # 1: 8-bit exact image set, the weak update reads the byte planes
    ctx.set_cameras(cams)
    p1 = wl.first_init_params(S, iters)
    ctx.set_params(p1)
    ctx.set_seed(1234 + rank)
    ctx.upload_state(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32), weak=np.full(L, synth.STRONG, np.uint8),
                     edge=edge, label=label, radius=np.full(L, 5, np.int32))
    del imgs
    weak_frac = 0.0
    if cfg["refine"]:
        # untimed FIRST_INIT pass -> hand-over -> the REFINE_ITER pass that is timed
        ctx.run_patchmatch()
        planes, views, weak, radius = ctx.download_state()
        st = wl.hand_over(planes, views, weak, radius, p1, W, H, extra_weak=wl.weak_mask(args.weak_layout, W, H, weak_frac_arg, flat))
        del planes, views, weak, radius
        p2 = wl.refine_iter_params(S, iters)
        for kv in (args.param if primary else []):
            k, v = kv.split("=")
            p2[k] = float(v) if "." in v else int(v)
        global WEAK_PEAK_RADIUS
        WEAK_PEAK_RADIUS = int(p2["weak_peak_radius"])
        ctx.set_params(p2)
        ctx.set_depths_device([deps[i].data_ptr() for i in order], W)
        ctx.upload_state(planes=st[0], views=st[1], weak=st[2], radius=st[3])
        weak_frac = ctx.weak_count() / float(L)
        del st
    global WEAK_COUNT
    # the weak update's launch structure depends on the size of its launches: one site of black + red, or a colour each
    WEAK_COUNT = ctx.weak_count() if weak_one_launch_site() else ctx.weak_count() // 2
    del deps
    ctx.save_state()
    ctx.synchronize()
    ctx.timings(reset=True)
    torch.cuda.empty_cache()
    t_setup = time.time() - t_setup

    def one_step(view_index, profile=False):
        ctx.set_seed(1234 + view_index)
        ctx.set_profiling(profile)
        ctx.restore_state()
        ctx.run_patchmatch()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # ---- warm-up (the first warm-up step also counts NCC evaluations: deterministic per seed) -------
    evals = None
    for w in range(max(warmup, 1)):
        one_step(rank, profile=(w == 0))
        if w == 0:
            ctx.synchronize()
            evals = ctx.timings(reset=True)
    ctx.set_profiling(False)
    ctx.synchronize()
    ctx.timings(reset=True)

    # ---- timed region ------------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for s in range(steps):
        one_step(rank + s * world)
    ctx.synchronize()
    busy = time.perf_counter() - t0          # this rank's own time (before waiting for the others)
    barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    busy_all = [busy]
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        bt = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(bt, torch.tensor([busy], dtype=torch.float64, device=dev))
        busy_all = [float(b.item()) for b in bt]

    if rank == 0:
        total_px_iter = float(W) * H * iters * steps * world
        value = total_px_iter / dt / 1e6
        stage_ms = {k: v for k, v in tm["stage_ms"].items() if v > 0}
        per_launch = {k: stage_ms[k] / max(tm["stage_launches"][k], 1) for k in stage_ms}
        ev_launch = {k: evals["ncc_evals"][k] / max(evals["stage_launches"][k], 1) for k in stage_ms}
        ranked = sorted((k for k in stage_ms if k != "strong_prep"), key=lambda k: -stage_ms[k])
        roofs = {k: roofline_of(k, S, W, H, per_launch[k], ev_launch[k]) for k in ranked[:4]}
        # what a WEAK pixel costs against any pixel: the launch sites that only touch WEAK pixels (anchor search, fit plane, weak
        # update) per WEAK pixel, everything else per pixel of the view (the regime must be visible in the line)
        weak_sites = ("find_nearest_strong", "gen_neighbours", "neighbour_update", "ransac_fit", "weak_update")
        weak_ms = sum(stage_ms.get(k, 0.0) for k in weak_sites) / steps
        other_ms = dt * 1e3 / steps - weak_ms
        n_weak = weak_frac * L
        dom = ranked[0]
        workload = "BASELINE %s stand-in: %dx%d, S=%d source views, %d PatchMatch iterations, %s" % (
            cfg_name if (W, H, S) == (cfg["W"], cfg["H"], cfg["S"]) else "custom(%s-like)" % cfg_name, W, H, S, iters,
            ("REFINE_ITER pass, geom_consistency on, use_APD on (%.1f %% WEAK pixels%s), edge/label/radius priors on, inputs from an untimed FIRST_INIT pass" % (100 * weak_frac, ", a few large connected regions" if args.weak_layout == "regions" else ""))
            if cfg["refine"] else "FIRST_INIT, geom off") + ((" [params overridden: %s]" % ", ".join(args.param)) if (primary and args.param) else "")
        out = {
            "metric": "Mpixels/sec/PatchMatch-iteration", "value": round(value, 3), "unit": "Mpx/s/iter",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic (procedural texture quantised to 8-bit grey levels, as decoded image files are; image_format=%d; camera rig: %s)" % (IMAGE_FORMAT, "per-view rotations 5-30 deg, per-view K" if args.rig == "rotated" else "R = I, one K (round-1/2 rig)"),
            "config": {"workload": workload + ", one reference view per step per GPU", "src_depths": args.src_depths if cfg["refine"] else None, "baseline_config": cfg_name,
                       "width": W, "height": H, "src_views": S, "iterations": iters, "weak_fraction": round(weak_frac, 4), "weak_layout": args.weak_layout,
                       "parallelism": "rank r takes view r mod %d of the scene as its reference view; %d rank(s), no data-path collective" % (NI, world)},
            "roofline": dict(roofs[dom], launch_site=dom, share_of_step=round(stage_ms[dom] / (dt * 1e3), 3)),
            "rooflines_top_kernels": roofs,
            "iter_loop_value": round(float(W) * H * iters * steps / (tm["iter_loop_ms"] * 1e-3) / 1e6, 3) if tm["iter_loop_ms"] > 0 else None,
            "stage_ms_per_step": {k: round(v / steps, 3) for k, v in stage_ms.items()},
            "ns_per_weak_pixel": round(weak_ms * 1e6 / n_weak, 1) if n_weak > 0 else None,
            "ns_per_pixel_other": round(other_ms * 1e6 / L, 1),
            # every kernel of a launch site (the PMC tooling averages the last `n` dispatches of each)
            "launches_per_step": dict([(kernel_name(k, S), primary_launches(k) * tm["stage_launches"][k] // steps) for k in stage_ms if k != "strong_prep"] +
                                      [(extra, mult * tm["stage_launches"][k] // steps) for k in stage_ms for extra, mult in extra_kernels(k, S)] +
                                      [(extra, tm["stage_launches"][k] // steps) for k, extra in
                                       (("gen_edge_inform", "dvp_gen_candidates_views"), ("strong_prep", "dvp_strong_search")) if k in stage_ms] +
                                      ([("dvp_weak_anchor_table", 1)] if ("weak_update" in stage_ms and os.environ.get("DVP_WEAK_ANCHOR_TAB", "1") != "0") else [])),
            "ncc_evals_per_step": {k: int(v) for k, v in evals["ncc_evals"].items() if v > 0},
            "Gevals_per_s": {k: round(evals["ncc_evals"][k] / (tm["stage_ms"][k] / steps * 1e-3) / 1e9, 3)
                             for k in evals["ncc_evals"] if evals["ncc_evals"][k] > 0 and tm["stage_ms"][k] > 0},
            "library_build_id": LOADED_BUILD_ID,
            "rank_busy_ms_per_step": [round(b / steps * 1e3, 1) for b in busy_all],
            "setup_s": round(t_setup, 1),
        }
        out["stage_ms_per_iteration"] = None
        if world == 1 and not args.no_per_iteration:
            out["stage_ms_per_iteration"] = per_iteration_ms(ctx, iters, rank, bool(cfg["refine"]) and weak_frac > 0)
        ctx.close()
        return out
    ctx.close()
    return None


def main():
    args = parse()
    if args.weak_frac is None:
        args.weak_frac = 0.10 if args.config == "cfg5" else 0.05
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)       # does not return
    # stdout carries ONE JSON line (rank 0).  Libraries write there too (RCCL prints its NCCL_DEBUG=VERSION banner and
    # its warnings on stdout): from here on file descriptor 1 is stderr, the JSON line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; n_gpus must be what was asked for" % (args.gpus, world))
    if args.dry_launch:
        return dry_launch(args, json_fd, rank, local_rank, world)
    import torch   # device plumbing + torch.distributed (RCCL); loaded first so one HIP runtime is shared
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # DVP_BENCH_FORCE_DIST=1 exercises the RCCL path (init, broadcast, all_reduce) with one rank
    use_dist = world > 1 or os.environ.get("DVP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pkg = importlib.import_module("dvp-mvs_amd")
    importlib.import_module("dvp-mvs_amd.workloads")
    wl = pkg.workloads
    cfg = dict(wl.CONFIGS[args.config])
    W, H = args.width or cfg["W"], args.height or cfg["H"]
    S, iters = args.src or cfg["S"], args.iters or cfg["iters"]
    env = dict(torch=torch, dist=dist, dev=dev, rank=rank, local_rank=local_rank, world=world, use_dist=use_dist, pkg=pkg)
    out = measure(env, args, args.config, W, H, S, iters, args.weak_frac, args.steps, args.warmup, primary=True)
    if rank == 0:
        # The other single-GPU configurations in the same line (the driver's one command observes cfg3 only):
        # a few steps each of cfg5 (>= 10 % WEAK as SURVEY 8d specifies) and cfg2, after the timed region of the primary one.
        default_workload = args.config == "cfg3" and (W, H, S, iters) == (cfg["W"], cfg["H"], cfg["S"], cfg["iters"]) and args.rig == "rotated"
        if world == 1 and default_workload and not args.no_secondary:
            sec = {}
            for name, wf in (("cfg5", 0.10), ("cfg2", 0.0)):
                c2 = wl.CONFIGS[name]
                o2 = measure(env, args, name, c2["W"], c2["H"], c2["S"], c2["iters"], wf, 5, 1, primary=False)
                sec[name] = {k: o2[k] for k in ("value", "ms_per_step", "steps", "warmup", "stage_ms_per_step", "ns_per_weak_pixel", "ns_per_pixel_other", "stage_ms_per_iteration")}
                sec[name].update(workload=o2["config"]["workload"], weak_fraction=o2["config"]["weak_fraction"],
                                 roofline={k: o2["roofline"].get(k) for k in ("launch_site", "kernel", "avg_launch_ms", "bound", "frac", "achieved", "peak", "unit", "traffic", "lane_utilisation", "Gevals_per_s", "useful_eval_rate_frac", "share_of_step")})
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, args, cfg, S, iters, device=local_rank)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
