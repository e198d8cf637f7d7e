#!/usr/bin/env python3
"""bench.py — Mpixels/s/PatchMatch-iteration of the MI355X PatchMatch engine.

A "step" = one APD::RunPatchMatch (reference: /root/reference/APD.cu:4406-4532) over one
reference view of synthetic input; the workload at N=1 is BASELINE.json configs[1]
("half-res, 5 source views, 6 iterations, 1xMI355X"): 3104x2064, S=5, 6 iterations, FIRST_INIT,
geom off.  With N>1 (torchrun, one rank per GPU) every rank processes its own views — weak
scaling, no data-path collective; the shared image/camera buffers are broadcast once over RCCL
before the timed region.  value = W*H*iters*steps*N / wall time of the timed steps (whole
RunPatchMatch, inputs resident in HBM).

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

NCC_BYTES = 724          # algorithmic bytes of one bilateral-NCC evaluation
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3104)
    ap.add_argument("--height", type=int, default=2064)
    ap.add_argument("--src", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=str, default="auto", help="WxH of the CPU-baseline view (auto: scaled to the core count)")
    ap.add_argument("--micro", action="store_true", help="also time the stand-alone cost-vector kernel")
    return ap.parse_args()


def cpu_baseline(synth, args, S, iters, capi=None, device=0):
    """The oracle ("port") timed on this host's cores on a bounded sample of the same workload:
    same scene generator, same S / iterations / params, a 256x192 view."""
    from oracle import oracle as O
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    if args.cpu_size == "auto":
        # ~0.004 Mpx*iter/s/core: ~2000 px per core keeps the sample at 10-30 s and
        # gives every OpenMP thread a few rows
        px = max(192 * 144, 2000 * ncores)
        h = int(round((px * 3 / 4) ** 0.5 / 8)) * 8
        w = h * 4 // 3
    else:
        w, h = [int(v) for v in args.cpu_size.split("x")]
    sc = synth.make_scene(w, h, S)
    p = bench_params(synth, S, iters)
    o = O.from_scene(sc, p)
    o.upload_state(planes=np.zeros((w * h, 4), np.float32), edge=sc["edge"], label=sc["label"],
                   radius=np.full(w * h, 5, np.int32))
    t0 = time.time()
    o.run_patchmatch()
    dt = time.time() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    omp = os.environ.get("OMP_NUM_THREADS")
    if omp:
        cores = min(cores, int(omp))
    res = {"value": round(w * h * iters / dt / 1e6, 5), "unit": "Mpx/s/iter", "cores": cores, "kind": "port",
           "sample": "%dx%d view, S=%d, %d iters, whole RunPatchMatch, oracle/ (OpenMP over rows), %.1f s" % (w, h, S, iters, dt)}
    if capi is not None:
        # parity of the engine on the very sample the CPU was timed on (oracle used as the checker, outside any timed region)
        g = capi.from_scene(sc, p, device=device)
        g.upload_state(planes=np.zeros((w * h, 4), np.float32), edge=sc["edge"], label=sc["label"], radius=np.full(w * h, 5, np.int32))
        g.run_patchmatch()
        a, b = o.get("planes"), g.get("planes")
        diff = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
        res["gpu_vs_cpu_plane_words_differing"] = int(diff.sum())
        res["gpu_vs_cpu_states_differing"] = int((o.get("weak_info") != g.get("weak_info")).sum() + (o.get("selected_views") != g.get("selected_views")).sum())
        g.close()
    return res


def bench_params(synth, S, iters):
    p = synth.default_params(S + 1, max_iterations=iters, state=synth.FIRST_INIT, use_APD=0)
    p["depth_min"] = np.float32(2.5) * np.float32(0.6)   # APD.cpp:1109-1110
    p["depth_max"] = np.float32(6.5) * np.float32(1.2)
    return p


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch   # device plumbing + torch.distributed (RCCL); loaded first so one HIP runtime is shared
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # DVP_BENCH_FORCE_DIST=1 exercises the RCCL path (init, broadcast, all_reduce) with one rank
    use_dist = world > 1 or os.environ.get("DVP_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    pkg = importlib.import_module("dvp-mvs_amd")
    synth, capi = pkg.synth, pkg.get_capi()
    W, H, S, iters = args.width, args.height, args.src, args.iters
    NI = S + 1

    # ---- inputs: rank 0 renders the scene; every rank gets it over RCCL (xGMI) --------------------
    pitch = W
    dev_imgs = torch.empty((NI, H, W), dtype=torch.float32, device="cuda")
    cams_t = torch.empty(NI * 112, dtype=torch.uint8, device="cuda")
    edge_t = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    if rank == 0:
        sc = synth.make_scene(W, H, S)
        dev_imgs.copy_(torch.from_numpy(sc["images"]))
        cams_t.copy_(torch.from_numpy(np.frombuffer(sc["cameras"].tobytes(), np.uint8).copy()))
        edge_t.copy_(torch.from_numpy(sc["edge"]))
    if use_dist:
        dist.broadcast(dev_imgs, 0)
        dist.broadcast(cams_t, 0)
        dist.broadcast(edge_t, 0)
    torch.cuda.synchronize()
    cams = np.frombuffer(cams_t.cpu().numpy().tobytes(), dtype=synth.CAMERA_DTYPE).copy()
    edge = edge_t.cpu().numpy()

    ctx = capi.Context(W, H, NI, device=local_rank)
    ctx.set_images_device([dev_imgs[i].data_ptr() for i in range(NI)], pitch)
    ctx.set_cameras(cams)
    ctx.set_params(bench_params(synth, S, iters))
    ctx.upload_state(edge=edge)
    del dev_imgs

    def one_step(view_index, profile=False):
        ctx.set_seed(1234 + view_index)
        ctx.set_profiling(profile)
        ctx.reset_state()
        ctx.run_patchmatch()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    # ---- warm-up (first warm-up step also counts NCC evaluations: deterministic per seed) ---------
    evals = None
    for w in range(max(args.warmup, 1)):
        one_step(rank, profile=(w == 0))
        if w == 0:
            ctx.synchronize()
            evals = ctx.timings(reset=True)
    ctx.set_profiling(False)
    ctx.synchronize()
    ctx.timings(reset=True)

    # ---- timed region ---------------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(rank + s * world)
    barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_px_iter = float(W) * H * iters * args.steps * world
        value = total_px_iter / dt / 1e6
        # dominant kernel: the strong red/black update (dvp_strong_update; the _v8 instantiation when S <= 8)
        launches = tm["stage_launches"]["strong_update"]
        avg_ms = tm["stage_ms"]["strong_update"] / max(launches, 1)
        ev_launch = evals["ncc_evals"]["strong_update"] / max(evals["stage_launches"]["strong_update"], 1)
        achieved = ev_launch * NCC_BYTES / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_strong_update.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/sec/PatchMatch-iteration", "value": round(value, 3), "unit": "Mpx/s/iter",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE cfg2 stand-in: %dx%d, S=%d source views, %d PatchMatch iterations, FIRST_INIT, geom off, one view per step per GPU" % (W, H, S, iters),
                       "width": W, "height": H, "src_views": S, "iterations": iters, "parallelism": "views round-robin over %d rank(s)" % world},
            "roofline": {"bound": "hbm", "kernel": "dvp_strong_update_v8" if S <= 8 else "dvp_strong_update", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "evals_per_launch": int(ev_launch), "bytes_per_eval": NCC_BYTES, "avg_launch_ms": round(avg_ms, 3),
                         "launches": launches,
                         # traffic (PMC, profiles/pmc_strong_update.json) over the same launch time: what HBM + Infinity Cache really moved
                         "physical_gbs": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if (traffic and avg_ms > 0) else None,
                         "note": "achieved = cache-oblivious algorithmic bytes (724 B per NCC evaluation, SURVEY 8d) / launch time: "
                                 "frac > 1 means L1/L2/Infinity Cache serve the re-reads; physical_gbs is what the PMC counters saw"},
            "iter_loop_value": round(float(W) * H * iters * args.steps / (tm["iter_loop_ms"] * 1e-3) / 1e6, 3) if tm["iter_loop_ms"] > 0 else None,
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in tm["stage_ms"].items() if v > 0},
            "evals_per_px_iter_strong": round(evals["ncc_evals"]["strong_update"] / (float(W) * H * iters), 2),
            "ncc_evals_per_step": {k: int(v) for k, v in evals["ncc_evals"].items() if v > 0},
            "Gevals_per_s": {k: round(evals["ncc_evals"][k] / (tm["stage_ms"][k] / args.steps * 1e-3) / 1e9, 3)
                             for k in evals["ncc_evals"] if evals["ncc_evals"][k] > 0 and tm["stage_ms"][k] > 0},
        }
        if args.micro:
            ms, ev = ctx.bench_cost_kernel(3)
            out["micro_cost_kernel"] = {"ms": round(ms, 3), "evals": ev, "GBps_algorithmic": round(ev * NCC_BYTES / (ms * 1e-3) / 1e9, 1)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth, args, S, iters, capi=capi, device=local_rank)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
