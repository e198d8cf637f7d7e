#!/usr/bin/env python3
"""Several scenes (dense folders) over the GPUs of ONE node — BASELINE cfg4 ("13 scenes, 8 x MI355X").

Scenes are independent jobs; what couples the GPUs inside a scene is the per-pass exchange of depth maps (DESIGN.md
section 6), and a pass ends when its busiest rank does: V views on N ranks leave ceil(V / N) * N - V view slots empty in
every pass (10 views on 8 ranks: 6 of 16).  Two ways to fill them, both exact — a scene's result never depends on what
else runs (the driver tests compare the files):

  --mode node  (default)  every scene runs on all N GPUs (`apd --rank r --world N`, longest-predicted-first view table,
                          RCCL exchange), `--in-flight M` scenes at a time: a GPU whose rank waits for the other ranks'
                          maps of scene A runs its views of scene B meanwhile (two processes per GPU; two full-size
                          contexts are 220 of the 288 GB).  tools/scale_sim.py predicts 7.1x at 8 ranks for the cfg4 mix
                          with M = 2 against 5.8x with M = 1 — a prediction, the node has never been available.
  --mode scenes           one GPU per scene, scenes handed out longest-first from a queue (no exchange at all); loses to
                          `node` as soon as one scene is larger than the fair share (ETH3D: facade, 76 of 454 views).

Scenes are started longest first (predicted cost = views x pixels of the first image, read from the file header).
usage: run_scenes.py [--gpus N] [--gpu-map d0,d1,...] [--mode node|scenes] [--in-flight M] [--transport rccl|host] scene_folder ... [-- apd options]"""
import argparse
import os
import struct
import subprocess
import sys
import threading
import time

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APD = os.path.join(HERE, "dvp-mvs_amd", "apd")


def gpu_count():
    n = 0
    top = "/sys/class/kfd/kfd/topology/nodes"
    if os.path.isdir(top):
        for d in os.listdir(top):
            try:
                props = dict(ln.split()[:2] for ln in open(os.path.join(top, d, "properties")) if len(ln.split()) >= 2)
                if int(props.get("simd_count", 0)) > 0 and int(props.get("gfx_target_version", 0)) > 0:
                    n += 1
            except OSError:
                pass
    return n


def image_size(path):
    """(w, h) from a JPEG frame header or a PNM header (host/io.cpp: ImageFileSize)"""
    for p in (path, os.path.splitext(path)[0] + ".pgm", os.path.splitext(path)[0] + ".ppm"):
        if not os.path.exists(p):
            continue
        with open(p, "rb") as f:
            head = f.read(1 << 16)
        if head[:2] == b"\xff\xd8":
            i = 2
            while i + 9 < len(head):
                if head[i] != 0xFF:
                    i += 1
                    continue
                m = head[i + 1]
                if m == 0xFF:
                    i += 1
                    continue
                if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
                    i += 2
                    continue
                ln = struct.unpack(">H", head[i + 2:i + 4])[0]
                if 0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
                    h, w = struct.unpack(">HH", head[i + 5:i + 9])
                    return w, h
                i += 2 + ln
        elif head[:2] in (b"P5", b"P6"):
            tok = [t for ln in head[2:200].split(b"\n") if not ln.startswith(b"#") for t in ln.split()]
            return int(tok[0]), int(tok[1])
    return 1, 1


def scene_cost(folder):
    try:
        toks = open(os.path.join(folder, "pair.txt")).read().split()
        n = int(toks[0])
        first = int(toks[1])
    except (OSError, ValueError, IndexError):
        return 0
    w, h = image_size(os.path.join(folder, "images", "%08d.jpg" % first))
    return n * w * h


def run_scene(folder, gpus, apd_args, transport, log_dir, tag):
    """one scene on the GPUs `gpus` (one rank each); returns the first non-zero exit status"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n = len(gpus)
    job = "scenes%d-%s-%d" % (os.getpid(), tag, int(time.time() * 1000) % 100000000)
    procs = []
    for r, g in enumerate(gpus):
        e = dict(env)
        e["HIP_VISIBLE_DEVICES"] = str(g)
        cmd = [APD, folder, "0"] + (["--rank", str(r), "--world", str(n), "--job", job, "--transport", transport] if n > 1 else []) + apd_args
        out = open(os.path.join(log_dir, "%s_rank%d.log" % (tag, r)), "w")
        procs.append((subprocess.Popen(cmd, env=e, stdout=out, stderr=subprocess.STDOUT), out))
    rc = 0
    for p, out in procs:
        s = p.wait()
        out.close()
        if s != 0 and rc == 0:
            rc = s
    return rc


def main():
    argv = sys.argv[1:]
    apd_args = []
    if "--" in argv:
        k = argv.index("--")
        argv, apd_args = argv[:k], argv[k + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="+")
    ap.add_argument("--gpus", type=int, default=0)
    ap.add_argument("--mode", default="node", choices=["node", "scenes"])
    ap.add_argument("--in-flight", type=int, default=2)
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"])
    ap.add_argument("--log-dir", default="")
    ap.add_argument("--gpu-map", default="", help="device of rank 0,1,... (default: rank r on GPU r); several ranks on one device need --transport host (tests on a one-GPU box)")
    a = ap.parse_args(argv)
    n = a.gpus or gpu_count()
    if n < 1:
        raise SystemExit("run_scenes.py: no GPU found on this node")
    devices = [int(t) for t in a.gpu_map.split(",")] if a.gpu_map else list(range(n))
    if len(devices) != n:
        raise SystemExit("run_scenes.py: --gpu-map needs one device per rank (%d)" % n)
    if len(set(devices)) < n and a.transport != "host":
        raise SystemExit("run_scenes.py: two ranks on one device need --transport host (RCCL refuses them)")
    log_dir = a.log_dir or os.path.join(a.scenes[0], "..")
    os.makedirs(log_dir, exist_ok=True)
    queue = sorted(a.scenes, key=lambda s: -scene_cost(s))
    lock = threading.Lock()
    failed = []
    t0 = time.time()

    def worker(gpus, slot):
        while True:
            with lock:
                if not queue or failed:
                    return
                s = queue.pop(0)
            tag = os.path.basename(os.path.normpath(s))
            t = time.time()
            rc = run_scene(s, gpus, apd_args, a.transport, log_dir, tag)
            with lock:
                print("scene %s on GPU(s) %s: %.1f s, exit %d" % (tag, ",".join(map(str, gpus)), time.time() - t, rc), flush=True)
                if rc != 0:
                    failed.append((s, rc))

    if a.mode == "scenes":
        threads = [threading.Thread(target=worker, args=([devices[g]], g)) for g in range(n)]
    else:
        threads = [threading.Thread(target=worker, args=(list(devices), k)) for k in range(max(1, a.in_flight))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print("%d scene(s) in %.1f s on %d GPU(s), mode %s%s" % (len(a.scenes), time.time() - t0, n, a.mode, "" if a.mode == "scenes" else ", %d in flight" % a.in_flight))
    if failed:
        raise SystemExit("run_scenes.py: %s failed with exit status %d" % failed[0])


if __name__ == "__main__":
    main()
