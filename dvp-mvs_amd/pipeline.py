"""In-memory, view-sharded pass scheduler for one scene.

Reproduces one round of the reference's schedule (/root/reference/main.cpp:450-512) — a FIRST_INIT
(or REFINE_INIT) pass followed by `geom_passes` REFINE_ITER passes with geometric consistency —
for all views of a scene held in memory, with the views sharded round-robin over the ranks of a
`torch.distributed` group (one process per GPU; backend "nccl" == RCCL over xGMI, "gloo" on CPU):

  * shared read-only inputs (images, cameras) are broadcast once from rank 0;
  * there is NO collective inside a pass — views are independent (`sharding.views_for_rank`);
  * after every pass each rank's new depth maps are exchanged with ONE all-gather (4 B/px/view),
    because the next geom pass reads the neighbours' depths (APD.cpp:1147-1166 reads only depths of
    the source views); normals / weak maps / view masks stay rank-local.

The reference processes views sequentially and in place, so inside a geom pass view k already sees
the depths views < k wrote in the same pass (Gauss-Seidel across views); the sharded schedule
reads the previous pass's depths for every view (Jacobi) — deterministic and independent of the
number of ranks, which is what makes 1-rank and N-rank runs bit-identical.

The engine is injected (`make_engine(width, height, num_images) -> Context-like`): the product
passes `capi.Context` (HIP); the CPU tests pass the oracle as a stand-in to exercise the sharding
and exchange logic under gloo.
"""
import numpy as np

from . import sharding, synth


def _post_process(planes, views, weak, radius, params, nsrc, scale_size=1, clean_masks=True):
    """What ProcessProblem does with the kernel outputs before writing them (main.cpp:298-363):
    zero out-of-range depths (-> UNKNOWN) and fill small holes of each view's visibility mask
    (4-connected components of invisible pixels smaller than 20*(8/scale)^2 become visible)."""
    depth = planes[:, 3].copy()
    bad = (depth < params["depth_min"]) | (depth > params["depth_max"])
    depth[bad] = 0
    weak = weak.copy()
    weak[bad] = synth.UNKNOWN
    if clean_masks:
        from scipy import ndimage
        H, W = params["_H"], params["_W"]
        thr = 20 * (8 // scale_size) * (8 // scale_size)
        v2 = np.zeros_like(views)
        four = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        for i in range(nsrc):
            vis = ((views >> i) & 1).astype(bool).reshape(H, W)
            lab, n = ndimage.label(~vis, structure=four)
            if n:
                cnt = np.bincount(lab.ravel())
                big = cnt >= thr
                big[0] = False
                vis = ~big[lab]
            v2 |= (vis.reshape(-1).astype(np.uint32) << i)
        views = v2
    return depth, weak, views


class ScenePipeline:
    def __init__(self, make_engine, images, cameras, pairs, group=None, seed=1234, sampler=0, clean_masks=True):
        """images [NV,H,W] f32, cameras [NV] CAMERA_DTYPE, pairs[v] = list of source view ids.
        On ranks != 0 `images`/`cameras` may be None: they are broadcast from rank 0."""
        self.dist = None
        self.rank, self.world = 0, 1
        if group is not None:
            import torch.distributed as dist
            self.dist = dist
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.make_engine = make_engine
        self.seed, self.sampler, self.clean_masks = seed, sampler, clean_masks
        self.images, self.cameras, self.pairs = self._broadcast_inputs(images, cameras, pairs)
        self.NV, self.H, self.W = self.images.shape
        self.mine = sharding.views_for_rank(self.NV, self.rank, self.world)
        L = self.H * self.W
        self.depths = np.zeros((self.NV, self.H, self.W), np.float32)     # replicated after each pass
        self.state = {v: dict(planes=np.zeros((L, 4), np.float32), views=np.zeros(L, np.uint32),
                              weak=np.full(L, synth.STRONG, np.uint8), radius=np.full(L, 5, np.int32)) for v in self.mine}
        self.edges = {v: np.zeros(L, np.uint8) for v in self.mine}
        self.pass_index = 0
        self._engines = {}   # recycled engines by image count

    # ---- collectives -------------------------------------------------------------------------------
    def _device(self):
        import torch
        return "cuda" if (self.dist is not None and self.dist.get_backend() == "nccl") else "cpu"

    def _broadcast_inputs(self, images, cameras, pairs):
        if self.dist is None:
            return np.ascontiguousarray(images, np.float32), np.ascontiguousarray(cameras), pairs
        import torch
        obj = [None]
        if self.rank == 0:
            obj = [(images.shape, pairs)]
        self.dist.broadcast_object_list(obj, src=0)
        shape, pairs = obj[0]
        dev = self._device()
        t = torch.empty(shape, dtype=torch.float32, device=dev)
        c = torch.empty(shape[0] * synth.CAMERA_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            t.copy_(torch.from_numpy(np.ascontiguousarray(images, np.float32)))
            c.copy_(torch.from_numpy(np.frombuffer(np.ascontiguousarray(cameras).tobytes(), np.uint8).copy()))
        self.dist.broadcast(t, src=0)       # RCCL broadcast of the shared image buffers
        self.dist.broadcast(c, src=0)
        cams = np.frombuffer(c.cpu().numpy().tobytes(), dtype=synth.CAMERA_DTYPE).copy()
        return t.cpu().numpy(), cams, pairs

    def _allgather_depths(self, new_depths):
        """new_depths: {view: [H,W]} of this rank -> self.depths for every view (one all-gather)."""
        if self.dist is None:
            for v, d in new_depths.items():
                self.depths[v] = d
            return
        import torch
        slots = (self.NV + self.world - 1) // self.world
        dev = self._device()
        send = torch.zeros((slots, self.H, self.W), dtype=torch.float32, device=dev)
        for i, v in enumerate(self.mine):
            send[i].copy_(torch.from_numpy(new_depths[v]))
        recv = torch.empty((self.world * slots, self.H, self.W), dtype=torch.float32, device=dev)
        self.dist.all_gather_into_tensor(recv, send)
        recv = recv.cpu().numpy().reshape(self.world, slots, self.H, self.W)
        for r in range(self.world):
            for i, v in enumerate(sharding.views_for_rank(self.NV, r, self.world)):
                self.depths[v] = recv[r, i]

    # ---- one pass over this rank's views -------------------------------------------------------------
    def run_pass(self, state, iters=3, geom=False, weak_peak_radius=6, use_apd=False, **param_overrides):
        new_depths = {}
        for v in self.mine:
            src = list(self.pairs[v])
            order = [v] + src
            NI = len(order)
            p = synth.default_params(NI, max_iterations=iters, state=state, use_APD=int(use_apd),
                                     geom_consistency=int(geom), weak_peak_radius=weak_peak_radius, **param_overrides)
            p["depth_min"] = np.float32(self.cameras[v]["depth_min"]) * np.float32(0.6)    # APD.cpp:1109-1110
            p["depth_max"] = np.float32(self.cameras[v]["depth_max"]) * np.float32(1.2)
            # engines that can be reset on the device are recycled per image count (no re-allocation
            # of the ~30 device buffers per view); others (the CPU oracle in tests) are created fresh
            eng = self._engines.pop(NI, None)
            if eng is None:
                eng = self.make_engine(self.W, self.H, NI)
            else:
                eng.reset_state()
            eng.set_images(self.images[order])
            eng.set_cameras(self.cameras[order])
            eng.set_params(p)
            eng.set_sampler(self.sampler)
            eng.set_seed(self.seed + v * 1000003 + self.pass_index)   # same rule as host/APD.cpp
            if geom:
                eng.set_depths(self.depths[order])
            st = self.state[v]
            eng.upload_state(planes=st["planes"], views=st["views"], weak=st["weak"], edge=self.edges[v], radius=st["radius"])
            eng.run_patchmatch()
            planes, views, weak, radius = eng.get("planes"), eng.get("selected_views"), eng.get("weak_info"), eng.get("radius")
            pp = {"depth_min": p["depth_min"], "depth_max": p["depth_max"], "_H": self.H, "_W": self.W}
            depth, weak, views = _post_process(planes, views, weak, radius, pp, len(src), clean_masks=self.clean_masks)
            planes = planes.copy()
            planes[:, 3] = depth
            radius = radius.copy()
            radius[weak == synth.UNKNOWN] = 5   # APD.cpp:1663-1666
            self.state[v] = dict(planes=planes, views=views, weak=weak, radius=radius)
            new_depths[v] = depth.reshape(self.H, self.W)
            if hasattr(eng, "reset_state"):
                self._engines[NI] = eng
            elif hasattr(eng, "close"):
                eng.close()
        self._allgather_depths(new_depths)
        self.pass_index += 1

    def run_round(self, iters=3, geom_passes=3, first=True):
        """pass A + `geom_passes` REFINE_ITER passes (main.cpp:452-510, single scale)."""
        self.run_pass(synth.FIRST_INIT if first else synth.REFINE_INIT, iters=iters, geom=False,
                      weak_peak_radius=6, use_apd=not first)
        for j in range(geom_passes):
            self.run_pass(synth.REFINE_ITER, iters=iters, geom=True, weak_peak_radius=max(4 - 2 * j, 2), use_apd=not first)
        return self.state
