"""An INDEPENDENT float64 numpy reading of the reference's cost functions, written from the source
text of /root/reference/APD.cu (not from oracle/): ComputeHomography (:679-739),
ComputeCorrespondingPoint (:741-748), ComputeBilateralNCCOld (:1023-1113), ComputeBilateralNCCNew
(:835-1021), ComputeGeomConsistencyCost (:1218-1256) with Get3DPointonWorld_cu (:467-487) and
ProjectonCamera_cu (:489-499), and DepthToWeak's cost line + peak rule (:3892-4051).

TEST INFRASTRUCTURE.  It shares no code with the oracle or the engine and uses none of the
numerics contract (no batched reciprocal, no fmaf, no 8-bit sampler, libm exp, float64 throughout),
so agreement with the oracle to float32 accuracy on ROTATED cameras with per-view K is evidence that
the oracle reads R, t, c and K the way the reference's source does — the one thing the
oracle-vs-engine bit comparisons cannot show.  Scalar loops: use on a few hundred samples."""
import math

import numpy as np


def cam64(cam):
    return dict(K=cam["K"].astype(np.float64), R=cam["R"].astype(np.float64), t=cam["t"].astype(np.float64),
                c=cam["c"].astype(np.float64), width=int(cam["width"]), height=int(cam["height"]))


def homography(ref, src, pl):
    """APD.cu:679-739, written out index by index like the source (R is row-major R[3*r + c])."""
    R, t = ref["R"], ref["t"]
    S, u = src["R"], src["t"]
    ref_C = [-(R[0 + k] * t[0] + R[3 + k] * t[1] + R[6 + k] * t[2]) for k in range(3)]
    src_C = [-(S[0 + k] * u[0] + S[3 + k] * u[1] + S[6 + k] * u[2]) for k in range(3)]
    Rrel = [S[3 * (m // 3) + 0] * R[3 * (m % 3) + 0] + S[3 * (m // 3) + 1] * R[3 * (m % 3) + 1] + S[3 * (m // 3) + 2] * R[3 * (m % 3) + 2]
            for m in range(9)]
    Crel = [ref_C[k] - src_C[k] for k in range(3)]
    trel = [S[3 * r + 0] * Crel[0] + S[3 * r + 1] * Crel[1] + S[3 * r + 2] * Crel[2] for r in range(3)]
    n = (pl[0], pl[1], pl[2])
    H = [Rrel[3 * r + c] - trel[r] * n[c] / pl[3] for r in range(3) for c in range(3)]
    K = ref["K"]
    tmp = [0.0] * 9
    for r in range(3):
        tmp[3 * r + 0] = H[3 * r + 0] / K[0]
        tmp[3 * r + 1] = H[3 * r + 1] / K[4]
        tmp[3 * r + 2] = -H[3 * r + 0] * K[2] / K[0] - H[3 * r + 1] * K[5] / K[4] + H[3 * r + 2]
    Ks = src["K"]
    out = [Ks[0] * tmp[0] + Ks[2] * tmp[6], Ks[0] * tmp[1] + Ks[2] * tmp[7], Ks[0] * tmp[2] + Ks[2] * tmp[8],
           Ks[4] * tmp[3] + Ks[5] * tmp[6], Ks[4] * tmp[4] + Ks[5] * tmp[7], Ks[4] * tmp[5] + Ks[5] * tmp[8],
           Ks[8] * tmp[6], Ks[8] * tmp[7], Ks[8] * tmp[8]]
    return out


def corresponding_point(H, x, y):
    z = H[6] * x + H[7] * y + H[8]
    return (H[0] * x + H[1] * y + H[2]) / z, (H[3] * x + H[4] * y + H[5]) / z


def tex_point(img, ix, iy):
    """tex2D(img, ix + 0.5, iy + 0.5), integer ix, iy: the texel itself, clamp addressing."""
    h, w = img.shape
    return float(img[min(max(iy, 0), h - 1), min(max(ix, 0), w - 1)])


def tex_linear(img, x, y):
    """tex2D(img, x + 0.5, y + 0.5), cudaFilterModeLinear, clamp (APD.cpp:1501-1517): bilinear between
    the texel centres around pixel coordinate (x, y); exact fractions (the texture unit's 8-bit weights
    are third-party arithmetic: the oracle is compared with its sampler 1)."""
    h, w = img.shape
    x = min(max(x, -1.0), float(w))
    y = min(max(y, -1.0), float(h))
    i0, j0 = math.floor(x), math.floor(y)
    a, b = x - i0, y - j0
    t00, t10 = tex_point(img, i0, j0), tex_point(img, i0 + 1, j0)
    t01, t11 = tex_point(img, i0, j0 + 1), tex_point(img, i0 + 1, j0 + 1)
    return (1 - b) * ((1 - a) * t00 + a * t10) + b * ((1 - a) * t01 + a * t11)


def _ncc(sr, srr, ss, sss, srs, sw):
    """APD.cu:1091-1109 (and 975-994).  min/max are CUDA's: a NaN operand is dropped."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float64(1.0) / np.float64(sw)
        sr, srr, ss, sss, srs = sr * inv, srr * inv, ss * inv, sss * inv, srs * inv
        var_ref = srr - sr * sr
        var_src = sss - ss * ss
        if var_ref < 1e-5 or var_src < 1e-5:
            return 2.0
        v = 1.0 - (srs - sr * ss) / math.sqrt(var_ref * var_src) if var_ref * var_src >= 0 else float("nan")
    if v != v:
        return 2.0     # max(0, min(2, NaN)) with fminf/fmaxf semantics = 2
    return max(0.0, min(2.0, v))


def ncc_old(images, cams, x, y, v, pl, radius=5, increment=2, sigma_spatial=5.0, sigma_color=3.0):
    ref, src = cams[0], cams[v]
    H = homography(ref, src, pl)
    px, py = corresponding_point(H, x, y)
    if px >= src["width"] or px < 0.0 or py >= src["height"] or py < 0.0:
        return 2.0
    centre = tex_point(images[0], x, y)
    sr = srr = ss = sss = srs = sw = 0.0
    for i in range(-radius, radius + 1, increment):
        for j in range(-radius, radius + 1, increment):
            a = tex_point(images[0], x + i, y + j)
            qx, qy = corresponding_point(H, x + i, y + j)
            b = tex_linear(images[v], qx, qy)
            w = math.exp(-math.sqrt(i * i + j * j) / (2.0 * sigma_spatial * sigma_spatial) - abs(a - centre) / (2.0 * sigma_color * sigma_color))
            sr += w * a
            srr += w * a * a
            ss += w * b
            sss += w * b * b
            srs += w * a * b
            sw += w
    return _ncc(sr, srr, ss, sss, srs, sw)


_RING5 = [(-5, -5), (-5, 0), (-5, 5), (0, -5), (0, 5), (5, -5), (5, 0), (5, 5)]


def ncc_new(images, cams, x, y, v, pl, anchors, anchor_views, anchor_offsets, radius=5, increment=2, sigma_color=3.0):
    """APD.cu:835-1021 for a WEAK pixel.  anchors: 12 (x, y), anchors[0] = the pixel, (-1, -1) = absent;
    anchor_views[k]: selected_views word of anchor k; anchor_offsets[k]: the 8 candidate offsets of anchor k
    for THIS view."""
    W, Hh = cams[0]["width"], cams[0]["height"]
    ref, src = cams[0], cams[v]
    H = homography(ref, src, pl)
    px, py = corresponding_point(H, x, y)
    if px >= src["width"] or px < 0.0 or py >= src["height"] or py < 0.0:
        return 2.0
    centre = tex_point(images[0], x, y)
    centre_cost, strong_cost, strong_count = 0.0, 0.0, 0
    for k in range(12):
        ax, ay = anchors[k]
        if ax == -1 or ay == -1:
            continue
        qx, qy = corresponding_point(H, ax, ay)
        if qx < 0 or qy < 0 or qx >= W or qy >= Hh:
            if k != 0:
                if (anchor_views[k] >> (v - 1)) & 1:
                    strong_cost += 2.0
                    strong_count += 1
                continue
            return 2.0
        taps = []
        if k == 0:
            taps = [(i, j) for i in range(-radius, radius + 1, increment) for j in range(-radius, radius + 1, increment)]
        elif (anchor_views[k] >> (v - 1)) & 1:
            for m in range(9):
                i, j = (0, 0) if m == 8 else anchor_offsets[k][m]
                if i == 0 and j == 0 and m < 8:
                    i, j = _RING5[m]
                taps.append((int(i), int(j)))
        sr = srr = ss = sss = srs = sw = 0.0
        for i, j in taps:
            a = tex_point(images[0], ax + i, ay + j)
            tx, ty = corresponding_point(H, ax + i, ay + j)
            b = tex_linear(images[v], tx, ty)
            w = math.exp(-abs(a - centre) / (2.0 * sigma_color * sigma_color))
            sr += w * a
            srr += w * a * a
            ss += w * b
            sss += w * b * b
            srs += w * a * b
            sw += w
        c = _ncc(sr, srr, ss, sss, srs, sw)    # no taps: 0/0 -> NaN -> 2 (the anchor is charged the maximum)
        if k == 0:
            centre_cost = c
        else:
            strong_cost += c
            strong_count += 1
    if strong_count == 0:
        return centre_cost
    return 0.25 * centre_cost + 0.75 * min(strong_cost / strong_count, 2.0)


def point_on_world(x, y, depth, cam):
    """APD.cu:467-487: R^T (depth K^-1 p) + c, K without skew."""
    K, R, c = cam["K"], cam["R"], cam["c"]
    X = (depth * (x - K[2]) / K[0], depth * (y - K[5]) / K[4], depth)
    return tuple(R[0 + k] * X[0] + R[3 + k] * X[1] + R[6 + k] * X[2] + c[k] for k in range(3))


def project_on_camera(P, cam):
    """APD.cu:489-499: K (R X + t) with the full K (skew included)."""
    K, R, t = cam["K"], cam["R"], cam["t"]
    T = [R[3 * r] * P[0] + R[3 * r + 1] * P[1] + R[3 * r + 2] * P[2] + t[r] for r in range(3)]
    d = K[6] * T[0] + K[7] * T[1] + K[8] * T[2]
    return (K[0] * T[0] + K[1] * T[1] + K[2] * T[2]) / d, (K[3] * T[0] + K[4] * T[1] + K[5] * T[2]) / d, d


def depth_from_plane(cam, pl, x, y):
    K = cam["K"]   # APD.cu:419-422
    return -pl[3] * K[0] / ((x - K[2]) * pl[0] + (K[0] / K[4]) * (y - K[5]) * pl[1] + K[0] * pl[2])


def distance_to_origin(cam, x, y, depth, n):
    K = cam["K"]   # APD.cu:400-405
    X = (depth * (x - K[2]) / K[0], depth * (y - K[5]) / K[4], depth)
    return -(n[0] * X[0] + n[1] * X[1] + n[2] * X[2])


def geom_cost(depth_maps, cams, x, y, v, pl):
    """APD.cu:1218-1256."""
    ref, src = cams[0], cams[v]
    z = depth_from_plane(ref, pl, x, y)
    fwd = point_on_world(x, y, z, ref)
    sx, sy, _ = project_on_camera(fwd, src)
    if not (math.isfinite(sx) and math.isfinite(sy)):
        return None
    zs = tex_point(depth_maps[v], int(sx), int(sy))      # (int) truncates toward zero
    if zs == 0.0:
        return 3.0
    back = point_on_world(sx, sy, zs, src)
    bx, by, _ = project_on_camera(back, ref)
    return min(3.0, math.hypot(x - bx, y - by))


def depth_to_weak(images, depth_maps, cams, x, y, plane_world, views, weights, depth_min, depth_max, geom, geom_factor=0.2,
                  weak_peak_radius=2, radius=5, increment=2):
    """APD.cu:3892-4051 for one pixel.  Returns (state, cost line [61], min_peak) with state 0 WEAK, 1 STRONG,
    2 UNKNOWN; None where the pixel sits within 1e-4 of a decision threshold (float32 vs float64 may differ)."""
    W, Hh = cams[0]["width"], cams[0]["height"]
    if x < 6 or y < 6 or x >= W - 6 or y >= Hh - 6:
        return 2, None, None
    R = cams[0]["R"]
    n = tuple(R[3 * r] * plane_world[0] + R[3 * r + 1] * plane_world[1] + R[3 * r + 2] * plane_world[2] for r in range(3))   # TransformNormal2RefCam
    z0 = float(plane_world[3])
    if z0 == 0:
        return 2, None, None
    S = len(cams) - 1
    sel = [s for s in range(1, S + 1) if (views >> (s - 1)) & 1]
    if not sel:
        return 2, None, None
    base = sum(math.sqrt(sum((cams[0]["c"][k] - cams[s]["c"][k]) ** 2 for k in range(3))) for s in sel) / len(sel)
    wn = float(sum(weights[s - 1] for s in sel))
    disp = cams[0]["K"][0] * base / z0
    line = []
    for pd in range(-30, 31):
        with np.errstate(divide="ignore"):
            z = float(np.float64(cams[0]["K"][0] * base) / np.float64(disp + pd))
        if z < depth_min or z > depth_max:
            line.append(2.0)
            continue
        pl = (n[0], n[1], n[2], distance_to_origin(cams[0], x, y, z, n))
        acc = 0.0
        for s in sel:
            c = ncc_old(images, cams, x, y, s, pl, radius, increment)
            if geom:
                c += geom_factor * geom_cost(depth_maps, cams, x, y, s, pl)
            acc += c * weights[s - 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            pc = float(np.float64(acc) / np.float64(wn))
        line.append(pc if 2.0 > pc else 2.0)     # MIN(2.0f, p_cost) = (2 > p ? p : 2): NaN -> 2
    eps = 1e-4
    fragile = False
    peaks = []
    min_peak, min_cost = 0, 2.0
    for i in range(2, 59):
        a, b, c = line[i - 1], line[i], line[i + 1]
        if 0 < abs(a - b) < eps or 0 < abs(c - b) < eps:     # (equal entries are saturated at 2.0 on both sides)
            fragile = True
        if a > b and c > b:
            peaks.append(i)
            if b < min_cost:
                if abs(b - min_cost) < eps:
                    fragile = True
                min_peak, min_cost = i, b
    for i in peaks:
        if i != min_peak and abs(line[i] - min_cost) < eps:
            fragile = True
    if abs(min_peak - 30) > weak_peak_radius or line[min_peak] > 0.5:
        st = 0
        if abs(line[min_peak] - 0.5) < eps:
            fragile = True
    elif len(peaks) == 1:
        st = 1 if line[min_peak] <= 0.15 else 0
        if abs(line[min_peak] - 0.15) < eps:
            fragile = True
    else:
        var = math.sqrt(sum((line[i] - min_cost) ** 2 for i in peaks if i != min_peak)) / (len(peaks) - 1)
        st = 1 if var > 0.2 else 0
        if abs(var - 0.2) < eps:
            fragile = True
    return (None if fragile else st), line, min_peak


def local_refine(images, depth_maps, cams, x, y, plane_world, views, weights, depth_min, depth_max, geom, geom_factor=0.2,
                 radius=5, increment=2):
    """APD.cu:4053-4139 for one pixel.  Returns (new depth or None = unchanged, fragile): fragile where the acceptance
    test or the choice between two sweep slots sits within 1e-4 of flipping (float32 vs float64)."""
    R = cams[0]["R"]
    n = tuple(R[3 * r] * plane_world[0] + R[3 * r + 1] * plane_world[1] + R[3 * r + 2] * plane_world[2] for r in range(3))   # TransformNormal2RefCam
    z0 = float(plane_world[3])
    if z0 == 0:
        return None, False
    S = len(cams) - 1
    sel = [s for s in range(1, S + 1) if (views >> (s - 1)) & 1]
    wn = float(sum(weights[s - 1] for s in sel))
    if wn == 0 or not sel:
        return None, False
    base = sum(math.sqrt(sum((cams[0]["c"][k] - cams[s]["c"][k]) ** 2 for k in range(3))) for s in sel) / len(sel)

    def total(z, separate):
        pl = (n[0], n[1], n[2], distance_to_origin(cams[0], x, y, z, n))
        acc = 0.0
        for s in sel:
            c = ncc_old(images, cams, x, y, s, pl, radius, increment)
            g = geom_factor * geom_cost(depth_maps, cams, x, y, s, pl) if geom else 0.0
            acc += (c * weights[s - 1] + g * weights[s - 1]) if separate else (c + g) * weights[s - 1]
        return acc / wn

    cost_now = total(z0, False)                       # :4085-4089: (ncc + factor * geom) * weight
    disp = cams[0]["K"][0] * base / z0
    eps = 1e-4
    fragile = False
    min_cost, best = 2.0, z0
    for pd in range(-5, 6):
        with np.errstate(divide="ignore"):
            z = float(np.float64(cams[0]["K"][0] * base) / np.float64(disp + pd))
        if z < depth_min or z > depth_max:
            continue
        t = total(z, True)                            # :4124-4126: ncc * weight and factor * geom * weight added one after the other
        if abs(t - min_cost) < eps:
            fragile = True
        if t < min_cost:
            min_cost, best = t, z
    if abs(cost_now - min_cost - 0.1) < eps:
        fragile = True
    return (best if cost_now - min_cost > 0.1 else None), fragile


def filter_strong(depth, costs, weak, W, H, x, y, STRONG=1):
    """CheckerboardFilterStrong (APD.cu:3184-3294) for one pixel on float32 arrays: the new depth, or None = untouched."""
    c = y * W + x
    if costs[c] < np.float32(0.001):
        return None
    taps = [(0, -1, y > 0), (0, -3, y > 2), (0, -5, y > 4), (0, 1, y < H - 1), (0, 3, y < H - 3), (0, 5, y < H - 5),
            (-1, 0, x > 0), (-3, 0, x > 2), (-5, 0, x > 4), (1, 0, x < W - 1), (3, 0, x < W - 3), (5, 0, x < W - 5),
            (2, -1, y > 0 and x < W - 2), (2, 1, y < H - 1 and x < W - 2), (-2, -1, y > 0 and x > 1), (-2, 1, y < H - 1 and x > 1),
            (-1, -2, x > 0 and y > 2), (1, -2, x < W - 1 and y > 2), (-1, 2, x > 0 and y < H - 2), (1, 2, x < W - 1 and y < H - 2)]
    vals = [depth[c]]
    for dx, dy, ok in taps:
        if ok and weak[(y + dy) * W + x + dx] == STRONG:
            vals.append(depth[(y + dy) * W + x + dx])
    vals = np.sort(np.array(vals, np.float32))
    m = len(vals) // 2
    return np.float32((vals[m - 1] + vals[m]) / np.float32(2)) if len(vals) % 2 == 0 else vals[m]


def get_depth_normal(cam, pl, x, y):
    """GetDepthandNormal (APD.cu:3167-3182): depth of the plane at the pixel, normal rotated to the world (TransformNormal)."""
    z = depth_from_plane(cam, pl, x, y)
    R = cam["R"]
    return (R[0] * pl[0] + R[3] * pl[1] + R[6] * pl[2], R[1] * pl[0] + R[4] * pl[1] + R[7] * pl[2], R[2] * pl[0] + R[5] * pl[1] + R[8] * pl[2], z)


def find_nearest_strong(weak, W, H, x, y, STRONG=1, max_radius=100):
    """FindNearestStrongPoint (APD.cu:4159-4193): rings of growing Chebyshev radius, x outer, y inner, first STRONG pixel."""
    for r in range(max_radius + 1):
        for dx in range(-r, r + 1):
            for dy in range(-r, r + 1):
                if abs(dx) != r and abs(dy) != r:
                    continue
                qx, qy = x + dx, y + dy
                if qx < 0 or qy < 0 or qx >= W or qy >= H:
                    continue
                if weak[qy * W + qx] == STRONG:
                    return qx, qy
    return -1, -1


def strong_propagation(images, cams, x, y, planes, costs, edge, edge_neigh, selected_views, W, H, it, uniforms, depth_min, depth_max,
                       radius=5, increment=2):
    """CheckerboardPropagationStrong, use_edge branch, up to the adoption of the best propagated plane (APD.cu:2010-2141,
    2462-2567), for one pixel.  planes / costs: the buffers as they stand BEFORE the launch (the contract's snapshot),
    camera-frame planes (n, offset).  uniforms: the 15 numbers of curand_uniform's place for this pixel and iteration.
    Returns dict(view_weight[S], selected (or None = not adopted), adopted position or None, fragile)."""
    S = len(cams) - 1
    f32 = np.float32
    eps = 2e-4
    fragile = False
    cost_array = [[0.0] * S for _ in range(8)]
    cost_array[0][0] = 2.0          # `= { 2.0f }` sets one element (APD.cu:2032)
    flag = [False] * 8
    positions = [0] * 8
    dirs = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (1, 1), (-1, 1), (1, -1)]
    c = y * W + x
    max_edge_dist = f32(max(H, W)) / f32(30.0)
    min_step = 2

    def vector(pos):
        pl = planes[pos].astype(np.float64)
        return [ncc_old(images, cams, x, y, v, pl, radius, increment) for v in range(1, S + 1)]

    def scan(di, step_num, step_len):
        dx, dy = dirs[di]
        best, best_cost = None, None
        for step in range(step_num):
            fx = fy = 0
            if di > 4:                    # (`> 4`, not `>= 4`: the source's test)
                if di % 2:
                    fx = dx
                else:
                    fy = dy
            tx, ty = x + 5 * dx + step * step_len * dx + fx, y + 5 * dy + step * step_len * dy + fy
            if not (0 <= tx < W and 0 <= ty < H):
                continue
            cst = costs[tx + ty * W]
            if best_cost is None or best_cost > cst:       # min_cost starts at FLT_MAX; `min_cost > cost`: a NaN never wins
                if cst == cst and cst < np.finfo(np.float32).max:
                    best, best_cost = tx + ty * W, cst
        return best

    for di in range(8):
        ex, ey = int(edge_neigh[c, di, 0]), int(edge_neigh[c, di, 1])
        dist = f32(math.sqrt(float(ex - x) ** 2 + float(ey - y) ** 2))
        if di >= 4:
            dist = f32(np.float64(dist) / math.sqrt(2.0))
        if edge[c]:
            dist = f32(11 * min_step)
        elif ey == -1 or dist >= max_edge_dist:          # `!edge_pt.x == -1` is never true (APD.cu:2059)
            dist = max_edge_dist
            if di >= 4:
                dist = f32(np.float64(dist) / math.sqrt(2.0))
        q = f32(f32(1.0) * dist / f32(min_step))
        if 0 < abs(float(q) - round(float(q))) < 1e-4:      # (an exact integer is exact in binary32 too)
            fragile = True
        step_num = min(max(11, int(q)), 22)
        q2 = f32(f32(1.0) * dist / f32(step_num))
        if 0 < abs(float(q2) - round(float(q2))) < 1e-4:
            fragile = True
        step_len = max(int(q2), min_step)
        if di < 4 and step_len % 2 == 1:
            step_len -= 1
        pos = scan(di, step_num, step_len)
        if pos is not None:
            flag[di] = True
            positions[di] = pos
            cost_array[di] = vector(pos)
    good_thr = 0.8 * math.exp(it * it / -90.0)
    if not edge[c]:
        for di in range(8):
            had = flag[di]
            pos = scan(di, 11, min_step)
            if pos is None:
                continue
            flag[di] = True
            tv = vector(pos)
            for val in cost_array[di][:S] + tv:
                if abs(val - good_thr) < eps or abs(val - 1.2) < eps:
                    fragile = True
            g0 = sum(v < good_thr for v in cost_array[di][:S]); b0 = sum(v > 1.2 for v in cost_array[di][:S])
            g1 = sum(v < good_thr for v in tv); b1 = sum(v > 1.2 for v in tv)
            if (not had) or g1 > g0 or (g1 == g0 and b1 < b0):
                positions[di] = pos
                cost_array[di] = tv
    # joint view selection (APD.cu:2462-2530)
    priors = [0.0] * S
    for i, nb in enumerate((c - W, c + W, c - 1, c + 1)):
        if flag[2 * i]:
            for j in range(S):
                priors[j] += 0.9 if (int(selected_views[nb]) >> j) & 1 else 0.1
    probs = [0.0] * S
    for i in range(S):
        count = cf = 0
        tmpw = 0.0
        for j in range(8):
            v = cost_array[j][i]
            if abs(v - good_thr) < eps or abs(v - 1.2) < eps:
                fragile = True
            if v < good_thr:
                tmpw += math.exp(v * v / -0.18)
                count += 1
            if v > 1.2:
                cf += 1
        if count > 2 and cf < 3:
            probs[i] = tmpw / count
        elif cf < 3:
            probs[i] = math.exp(good_thr * good_thr / -0.32)
        probs[i] *= priors[i]
    tot = sum(probs)
    vw = [0] * S
    if tot > 0:
        cdf, cum = [], 0.0
        for pr in probs:
            cum += pr / tot
            cdf.append(cum)
        for u in uniforms:
            rp = u - 1.1920929e-07
            for j in range(S):
                if abs(cdf[j] - rp) < 1e-5:
                    fragile = True
                if cdf[j] > rp:
                    vw[j] += 1
                    break
    else:
        fragile = True          # 1 / 0 and NaN comparisons: float32 territory
    wn = float(sum(vw))
    out = dict(view_weight=vw, selected=None, adopted=None, fragile=fragile, wn=wn, plane=None, cost=None)
    if wn == 0:
        return out
    final = [sum(vw[j] * cost_array[i][j] for j in range(S) if vw[j] > 0) / wn for i in range(8)]
    mi, mc = 0, final[0]
    for i in range(1, 8):                      # FindMinCostIndex: ties go to the LAST (APD.cu:155-166)
        if abs(final[i] - mc) < eps and final[i] != mc:
            out["fragile"] = True
        if final[i] <= mc:
            mc, mi = final[i], i
    now = vector(c)
    cost_now = sum(vw[j] * now[j] for j in range(S)) / wn
    out["plane"], out["cost"] = planes[c].astype(np.float64), cost_now
    if flag[mi]:
        z = depth_from_plane(cams[0], planes[positions[mi]].astype(np.float64), x, y)
        if abs(final[mi] - cost_now) < eps:
            out["fragile"] = True
        if depth_min <= z <= depth_max and final[mi] < cost_now:
            out["selected"] = sum(1 << j for j in range(S) if vw[j] > 0)
            out["adopted"] = positions[mi]
            out["plane"], out["cost"] = planes[positions[mi]].astype(np.float64), final[mi]
    return out


def _view_direction(cam, px, py, depth):
    """GetViewDirection (APD.cu:386-398)."""
    K = cam["K"]
    X = (depth * (px - K[2]) / K[0], depth * (py - K[5]) / K[4], depth)
    nrm = math.sqrt(X[0] ** 2 + X[1] ** 2 + X[2] ** 2)
    return (X[0] / nrm, X[1] / nrm, X[2] / nrm)


def random_normal_yzl(cams, x, y, depth, selected, uniforms, depth_maps=None):
    """GenerateRandomNormal_YZL (APD.cu:501-585) without the geometric term (src_depth = 1): the view directions of the
    reference and of the selected sources — the latter through the source's quirks: (x, y, x) as the direction vector and
    A[7] twice in the last row of matMul3x1 — then up to 200 rejection rounds of a uniform point on the sphere.
    uniforms: callable k -> the k-th number of the stream.  Returns (normal, fragile)."""
    dirs = [_view_direction(cams[0], x, y, depth)]
    Rr = cams[0]["R"]
    for s in range(1, len(cams)):
        if not (selected >> (s - 1)) & 1:
            continue
        fwd = point_on_world(x, y, depth, cams[0])
        sx, sy, _ = project_on_camera(fwd, cams[s])
        ix, iy = int(float(int(sx)) + 0.5), int(float(int(sy)) + 0.5)      # make_int2((int)x + 0.5f, ...)
        src_depth = 1.0                                                     # geom off; outside the image: uninitialised in the source, defined as 1
        if depth_maps is not None and 0 <= ix < cams[0]["width"] and 0 <= iy < cams[0]["height"]:
            src_depth = float(tex_point(depth_maps[s], int(sx), int(sy)))     # tex2D(depth_image, (int)x + 0.5f, (int)y + 0.5f)
        d = _view_direction(cams[s], ix, iy, src_depth)
        Rs = cams[s]["R"]
        Rt = [Rs[j * 3 + i] for i in range(3) for j in range(3)]            # transpose
        Rc = [sum(Rr[i * 3 + k] * Rt[k * 3 + j] for k in range(3)) for i in range(3) for j in range(3)]
        v = (d[0], d[1], d[0])                                              # {x, y, x} (APD.cu:543)
        f = (Rc[0] * v[0] + Rc[1] * v[1] + Rc[2] * v[2], Rc[3] * v[0] + Rc[4] * v[1] + Rc[5] * v[2], Rc[6] * v[0] + Rc[7] * v[1] + Rc[7] * v[2])   # A[7] twice (APD.cu:17)
        nrm = math.sqrt(f[0] ** 2 + f[1] ** 2 + f[2] ** 2)
        if len(dirs) < 20:
            dirs.append((f[0] / nrm, f[1] / nrm, f[2] / nrm))
    k = 0
    fragile = False
    n = (0.0, 0.0, 0.0)
    times = 200
    while times > 0:
        s_ = 2.0
        while s_ >= 1.0:
            q1 = 2.0 * uniforms(k) - 1.0
            q2 = 2.0 * uniforms(k + 1) - 1.0
            k += 2
            s_ = q1 * q1 + q2 * q2
            if abs(s_ - 1.0) < 1e-6:
                fragile = True
        sq = math.sqrt(1.0 - s_)
        n = (2.0 * q1 * sq, 2.0 * q2 * sq, 1.0 - 2.0 * s_)
        ok = True
        for dv in dirs:
            dot = n[0] * dv[0] + n[1] * dv[1] + n[2] * dv[2]
            if abs(dot) < 1e-6:
                fragile = True
            if dot > 0.0:
                ok = False
                break
        if ok:
            break
        times -= 1
    nrm = math.sqrt(n[0] ** 2 + n[1] ** 2 + n[2] ** 2)
    return (n[0] / nrm, n[1] / nrm, n[2] / nrm), fragile


def strong_refinement(images, cams, x, y, plane, cost, vw, wn, selected, depth_min, depth_max, u_depth, u_pert, u_normal, radius=5, increment=2):
    """PlaneHypothesisRefinementStrong (APD.cu:1311-1383): six hypotheses built from the values at entry, each adopted when its
    depth is in range and its weighted cost is below the running best.  Returns (plane, cost, fragile)."""
    S = len(cams) - 1
    depth = depth_from_plane(cams[0], plane, x, y)
    depth_rand = u_depth * (depth_max - depth_min) + depth_min
    n_rand, fragile = random_normal_yzl(cams, x, y, depth, selected, u_normal)
    lo, hi = (1 - 0.02) * depth, (1 + 0.02) * depth
    depth_pert = u_pert * (hi - lo) + lo
    nn = math.sqrt(plane[0] ** 2 + plane[1] ** 2 + plane[2] ** 2)
    n_pert = (plane[0] / nn, plane[1] / nn, plane[2] / nn)                  # GeneratePerturbedNormal returns the normalised input (APD.cu:617-661)
    n0 = (plane[0], plane[1], plane[2])
    hyps = [(depth_rand, n0), (depth, n_rand), (depth_rand, n_rand), (depth, n_pert), (depth, n_pert), (depth_pert, n0)]
    best_plane, best_cost = tuple(plane), cost
    for z, n in hyps:
        pl = (n[0], n[1], n[2], distance_to_origin(cams[0], x, y, z, n))
        t = sum(vw[j] * ncc_old(images, cams, x, y, j + 1, pl, radius, increment) for j in range(S) if vw[j] > 0) / wn
        zb = depth_from_plane(cams[0], pl, x, y)
        same = max(abs(a - b) / max(1e-2, abs(b)) for a, b in zip(pl, best_plane)) < 5e-5   # (the re-normalised current plane: either answer is the same plane)
        if abs(t - best_cost) < 2e-4 and not same:
            fragile = True
        if depth_min <= zb <= depth_max and t < best_cost:
            best_plane, best_cost = pl, t
    return best_plane, best_cost, fragile


def gen_edge_inform(image, selected_views, edge, label, weak, W, H, x, y, S, weak_radius=5, strong_radius=5, sigma_color=3.0, WEAK=0):
    """GenEdgeInform (APD.cu:3731-3890) for one pixel: the visibility-prior offsets per source view (the best-weighted
    neighbour that selected the view in each 30-degree sector, the sectors then ranked by that weight, the first eight
    kept), the nearest edge pixel in the eight directions, the edge density sigmoid of a WEAK pixel and — with a positive
    label — the farthest pixel of the own label before a label -1 in the eight directions.
    Returns dict(candidates[S][8] (or None where a ranking is fragile), edge_neigh[8], complex, label_boundary)."""
    c = y * W + x
    cands = []
    centre = float(image[y, x])
    for v in range(S):
        regions = [[] for _ in range(12)]
        for i in range(-weak_radius, weak_radius + 1):
            for j in range(-weak_radius, weak_radius + 1):
                if i == 0 and j == 0:
                    continue
                qx, qy = x + i, y + j
                if not (0 <= qx < W and 0 <= qy < H):
                    continue
                if not (int(selected_views[qx + qy * W]) >> v) & 1:
                    continue
                ang = math.degrees(math.atan2(float(j), float(i)))
                if ang < 0:
                    ang += 360.0
                ang = float(np.float32(ang))           # `float angle = calculateAngle(i, j)` (APD.cu:3759)
                w = math.exp(-abs(float(image[qy, qx]) - centre) / (2.0 * sigma_color * sigma_color))   # ComputeBilateralWeight_YZL: colour only
                r = int(ang // 30.0) if 0 <= ang < 360 else -1
                if len(regions[r]) < 20:               # Point regions[12][20] (APD.cu:3751)
                    regions[r].append((w, i, j))
        fragile = False
        heads = []
        for r in range(12):
            if not regions[r]:
                heads.append((0.0, 0, 0))              # regions[i][0] of an empty sector: defined as offset (0, 0), weight 0
                continue
            best = regions[r][0]
            for t in regions[r][1:]:                   # bubbleSort descending, swaps on strict `<`: the first of equals stays first
                if abs(t[0] - best[0]) < 1e-6 * max(best[0], 1e-30) and t[0] != best[0]:
                    fragile = True
                if t[0] > best[0]:
                    best = t
            heads.append(best)
        order = sorted(range(12), key=lambda r: -heads[r][0])      # stable: ties keep sector order, like the bubble sort
        for a in range(11):
            wa, wb = heads[order[a]][0], heads[order[a + 1]][0]
            if wa != wb and abs(wa - wb) < 1e-6 * max(wa, 1e-30):
                fragile = True
        cands.append(None if fragile else [(heads[r][1], heads[r][2]) for r in order[:8]])
    dirs = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (1, 1), (-1, 1), (1, -1)]
    en = []
    for dx, dy in dirs:
        nx, ny, hit = x + dx, y + dy, (-1, -1)
        while 0 <= nx < W and 0 <= ny < H:
            if edge[nx + ny * W]:
                hit = (nx, ny)
                break
            nx += dx
            ny += dy
        en.append(hit)
    out = dict(candidates=cands, edge_neigh=en, complex=None, label_boundary=None)
    if weak[c] == WEAK:
        ep = tot = 0
        for i in range(-strong_radius, strong_radius + 1):
            for j in range(-strong_radius, strong_radius + 1):
                nx, ny = x + i, y + j
                if 0 <= nx < W and 0 <= ny < H:
                    ep += int(edge[ny * W + nx] != 0)
                    tot += 1
        out["complex"] = 1.0 / (1.0 + math.exp(-25.0 * (ep / tot - 0.35)))
        if label[c] > 0:
            lb = []
            for dx, dy in dirs:
                nx, ny, last = x + dx, y + dy, (-1, -1)
                while 0 <= nx < W and 0 <= ny < H:
                    nl = label[nx + ny * W]
                    if nl == label[c]:
                        last = (nx, ny)
                    elif nl == -1:
                        break
                    nx += dx
                    ny += dy
                lb.append(last)
            out["label_boundary"] = lb
    return out


def bresenham_hits_edge(edge, W, H, A, B):
    """BresenhamLine(A, B) (APD.cu:267-311): walks from B towards A for at most max(H, W) / 30 steps; true when it meets an
    edge pixel.  False at once for long segments and for segments whose END POINTS lie on an edge."""
    max_step = int(max(H, W) / 30.0)
    x0, y0, x1, y1 = B[0], B[1], A[0], A[1]
    if (A[0] - B[0]) ** 2 + (A[1] - B[1]) ** 2 > 9 * max_step * max_step:
        return False
    if edge[x0 + y0 * W] or edge[x1 + y1 * W]:
        return False
    dx, sx = abs(x1 - x0), (1 if x0 < x1 else -1)
    dy, sy = abs(y1 - y0), (1 if y0 < y1 else -1)
    err = (dx if dx > dy else dy) // 2
    step = 0
    tagx = tagy = True
    while tagx or tagy:
        if x0 == x1:
            tagx = False
        if y0 == y1:
            tagy = False
        e2 = err
        if e2 > -dx:
            err -= dy
            x0 += sx
        if e2 < dy:
            err += dx
            y0 += sy
        if edge[x0 + y0 * W]:
            return True
        step += 1
        if step >= max_step:
            break
    return False


def _point_in_triangle(A, B, C, P):
    """PointinTriangle (APD.cu:244-265)."""
    ab = math.hypot(B[0] - A[0], B[1] - A[1]); bc = math.hypot(C[0] - B[0], C[1] - B[1]); ca = math.hypot(A[0] - C[0], A[1] - C[1])
    if ab <= 2 or bc <= 2 or ca <= 2:
        return False
    if not (ab + bc > ca and bc + ca > ab and ab + ca > bc):
        return False
    pa, pb, pc = (A[0] - P[0], A[1] - P[1]), (B[0] - P[0], B[1] - P[1]), (C[0] - P[0], C[1] - P[1])
    cr = lambda u, v: u[0] * v[1] - u[1] * v[0]
    t1, t2, t3 = cr(pa, pb), cr(pb, pc), cr(pc, pa)
    return t1 * t2 >= 0 and t1 * t3 >= 0


def ransac_fit_plane(cam, planes, anchors, edge, edge_neigh, label_boundary, W, H, x, y, draws, edge_limit, use_radius, use_edge, use_label, strong_radius=5):
    """RANSACToGetFitPlane (APD.cu:4195-4405) for one WEAK pixel.  anchors: neighbours[1..11] ((-1, -1) = none); planes:
    camera-frame planes; draws(k): the k-th 32-bit number of the pixel's stream.  Returns (fit plane or None = (0,0,0,0),
    radius or None = unchanged, fragile)."""
    pts = [a for a in anchors if a[0] != -1 and a[1] != -1]
    c = y * W + x
    if len(pts) < 3:
        return tuple(planes[c].astype(np.float64)), None, False
    K = cam["K"]
    P3, N = [], []
    for a in pts:
        pl = planes[a[0] + a[1] * W].astype(np.float64)
        z = depth_from_plane(cam, pl, a[0], a[1])
        P3.append((z * (a[0] - K[2]) / K[0], z * (a[1] - K[5]) / K[4], z))
        N.append(pl[:3])
    n = len(pts)
    cache = {}
    best, best_cost, tri = None, None, None
    fragile = False
    k = 0
    for _ in range(50):
        ia, ib, ic = draws(k) % n, draws(k + 1) % n, draws(k + 2) % n
        k += 3
        if ia == ib or ib == ic or ia == ic:
            continue
        dots = (float(np.dot(N[ia], N[ib])), float(np.dot(N[ia], N[ic])), float(np.dot(N[ib], N[ic])))
        if any(abs(d - 0.9) < 1e-5 for d in dots):
            fragile = True
        if dots[0] < 0.9 or dots[1] < 0.9 or dots[2] < 0.9:
            continue
        if not _point_in_triangle(pts[ia], pts[ib], pts[ic], (x, y)):
            continue
        if edge_limit:
            for u, v in ((ia, ib), (ib, ic), (ic, ia)):         # symmetric cache, the first asker's orientation (APD.cu:4260-4299)
                if (u, v) not in cache:
                    cache[(u, v)] = cache[(v, u)] = bresenham_hits_edge(edge, W, H, pts[u], pts[v])
            if cache[(ia, ib)] or cache[(ib, ic)] or cache[(ic, ia)]:
                continue
        A, B, C = P3[ia], P3[ib], P3[ic]
        ac = (A[0] - C[0], A[1] - C[1], A[2] - C[2]); bc = (B[0] - C[0], B[1] - C[1], B[2] - C[2])
        cv = (ac[1] * bc[2] - bc[1] * ac[2], -(ac[0] * bc[2] - bc[0] * ac[2]), ac[0] * bc[1] - bc[0] * ac[1])
        nn = math.sqrt(cv[0] ** 2 + cv[1] ** 2 + cv[2] ** 2)
        if nn == 0 or nn != nn:
            continue
        cv = (cv[0] / nn, cv[1] / nn, cv[2] / nn)
        w = -(cv[0] * A[0] + cv[1] * A[1] + cv[2] * A[2])
        cost = 0.0
        for si in range(n):
            if si in (ia, ib, ic):
                continue
            fx, fy = (pts[si][0] - K[2]) / K[0], (pts[si][1] - K[5]) / K[4]
            cost += abs(-w / (cv[0] * fx + cv[1] * fy + cv[2]) - P3[si][2])
        if best_cost is not None and abs(cost - best_cost) < 1e-5 * max(1.0, best_cost) and cost != best_cost:
            fragile = True
        if best_cost is None or cost < best_cost:
            best_cost, best, tri = cost, (cv[0], cv[1], cv[2], w), (ia, ib, ic)
    if best is None:
        return None, (strong_radius if use_radius else None), fragile
    z = depth_from_plane(cam, planes[c].astype(np.float64), x, y)
    vd = _view_direction(cam, x, y, z)
    if best[0] * vd[0] + best[1] * vd[1] + best[2] * vd[2] > 0:
        best = tuple(-t for t in best)
    radius = None
    if use_radius:
        A, B, C = pts[tri[0]], pts[tri[1]], pts[tri[2]]      # (use_a_index is never assigned in the source: defined as the winning triangle)
        a, b, c_ = math.hypot(A[0] - B[0], A[1] - B[1]), math.hypot(B[0] - C[0], B[1] - C[1]), math.hypot(C[0] - A[0], C[1] - A[1])
        p = (a + b + c_) / 2.0
        area = math.sqrt(max(0.0, p * (p - a) * (p - b) * (p - c_)))
        q = math.sqrt(area) / 2.0
        if abs(q - round(q)) < 1e-4:
            fragile = True
        radius = int(math.floor(q))
        md = min(math.hypot(A[0] - x, A[1] - y), math.hypot(B[0] - x, B[1] - y), math.hypot(C[0] - x, C[1] - y))
        if 2.5 * md < radius:
            radius = int(md)
        if edge_limit:
            if use_edge:
                ds = [math.hypot(e[0] - x, e[1] - y) for e in edge_neigh if e[0] != -1 and e[1] != -1]
                if ds and min(ds) < radius:
                    radius = int(min(ds))
            if use_label and label_boundary is not None:
                ds = [math.hypot(e[0] - x, e[1] - y) for e in label_boundary if e[0] != -1 and e[1] != -1]
                if ds and min(ds) < radius:
                    radius = int(min(ds))
        while (radius << 1) % 5 != 0:
            radius -= 1
        radius = 0 if radius < strong_radius else radius
    return best, radius, fragile


def weak_update(images, depth_maps, cams, x, y, planes, weak, selected_views, anchors, cand, fit_plane, W, H, it, u_view, u_depth, u_pert, u_normal,
                depth_min, depth_max, geom, geom_factor, radius, increment, STRONG=1):
    """CheckerboardPropagationWeak + PlaneHypothesisRefinementWeak (APD.cu:2739-3089, 1897-2008) for one WEAK pixel, REFINE_ITER
    write-back.  anchors: neighbours[0..11]; cand[pixel][view][8][2]: visibility-prior offsets.  Returns dict(view_weight,
    selected or None, plane, fragile).  (The launch's last step, the plain NCC of the final plane, is ncc_old.)"""
    S = len(cams) - 1
    eps = 2e-4
    fragile = False
    c = y * W + x
    av = [0 if a[0] == -1 else int(selected_views[a[0] + a[1] * W]) for a in anchors]

    def vector(pl):
        out = []
        for v in range(1, S + 1):
            offs = [None if a[0] == -1 else cand[a[0] + a[1] * W, v - 1] for a in anchors]
            out.append(ncc_new(images, cams, x, y, v, pl, anchors, av, offs, radius=radius, increment=increment))
        return out

    def with_geom(vec, pl, vw):
        return sum(vw[j] * (vec[j] + (geom_factor * geom_cost(depth_maps, cams, x, y, j + 1, pl) if geom else 0.0)) for j in range(S) if vw[j] > 0)

    cost_array = [[0.0] * S for _ in range(8)]
    cost_array[0][0] = 2.0
    flag, pos = [False] * 8, [0] * 8
    for i in range(8):
        a = anchors[i + 1]
        if a[0] == -1 or a[1] == -1 or weak[a[0] + a[1] * W] != STRONG:
            continue
        flag[i], pos[i] = True, a[0] + a[1] * W
        cost_array[i] = vector(planes[pos[i]].astype(np.float64))
    priors = [0.0] * S
    for i in range(8):
        a = anchors[i + 1]
        if a[0] == -1 or a[1] == -1:
            continue
        for j in range(S):
            priors[j] += 0.9 if (int(selected_views[a[0] + a[1] * W]) >> j) & 1 else 0.1
    thr = 0.8 * math.exp(it * it / -90.0)
    probs = [0.0] * S
    for i in range(S):
        count = cf = 0
        tmpw = 0.0
        for j in range(8):
            v = cost_array[j][i]
            if abs(v - thr) < eps or abs(v - 1.2) < eps:
                fragile = True
            if v < thr:
                tmpw += math.exp(v * v / -0.18)
                count += 1
            if v > 1.2:
                cf += 1
        if count > 2 and cf < 3:
            probs[i] = tmpw / count
        elif cf < 3:
            probs[i] = math.exp(thr * thr / -0.32)
        probs[i] *= priors[i]
    tot = sum(probs)
    vw = [0] * S
    out = dict(view_weight=vw, selected=None, plane=None, fragile=True)
    if not tot > 0:
        return out
    cdf, cum = [], 0.0
    for pr in probs:
        cum += pr / tot
        cdf.append(cum)
    for u in u_view:
        rp = u - 1.1920929e-07
        for j in range(S):
            if abs(cdf[j] - rp) < 1e-5:
                fragile = True
            if cdf[j] > rp:
                vw[j] += 1
                break
    wn = float(sum(vw))
    if wn == 0:
        return out
    final = []
    for i in range(8):
        if geom:
            t = sum(vw[j] * (cost_array[i][j] + geom_factor * (geom_cost(depth_maps, cams, x, y, j + 1, planes[pos[i]].astype(np.float64)) if flag[i] else 3.0))
                    for j in range(S) if vw[j] > 0)
        else:
            t = sum(vw[j] * cost_array[i][j] for j in range(S) if vw[j] > 0)
        final.append(t / wn)
    mi, mc = 0, final[0]
    for i in range(1, 8):
        if abs(final[i] - mc) < eps and final[i] != mc:
            fragile = True
        if final[i] <= mc:
            mc, mi = final[i], i
    plane = planes[c].astype(np.float64)
    cost = with_geom(vector(plane), plane, vw) / wn
    selected = None
    sel_live = int(selected_views[c])
    if flag[mi]:
        cand_pl = planes[pos[mi]].astype(np.float64)
        z = depth_from_plane(cams[0], cand_pl, x, y)
        if abs(final[mi] - cost) < eps:
            fragile = True
        if depth_min <= z <= depth_max and final[mi] < cost:
            plane, cost = cand_pl, final[mi]
            selected = sel_live = sum(1 << j for j in range(S) if vw[j] > 0)
    # PlaneHypothesisRefinementWeak
    fp = fit_plane.astype(np.float64)
    if not (fp[0] == 0 and fp[1] == 0 and fp[2] == 0):
        def consider(pl, plane, cost, fragile):
            t = with_geom(vector(pl), pl, vw) / wn
            zb = depth_from_plane(cams[0], pl, x, y)
            same = max(abs(a - b) / max(1e-2, abs(b)) for a, b in zip(pl, plane)) < 5e-5
            if abs(t - cost) < eps and not same:
                fragile = True
            if depth_min <= zb <= depth_max and t < cost:
                return tuple(pl), t, fragile
            return plane, cost, fragile
        plane, cost, fragile = consider(tuple(fp), tuple(plane), cost, fragile)
        depth = depth_from_plane(cams[0], plane, x, y)
        depth_rand = u_depth * (depth_max - depth_min) + depth_min
        n_rand, fr = random_normal_yzl(cams, x, y, depth, sel_live, u_normal, depth_maps if geom else None)
        fragile = fragile or fr
        lo, hi = (1 - 0.02) * depth, (1 + 0.02) * depth
        depth_pert = u_pert * (hi - lo) + lo
        nn = math.sqrt(plane[0] ** 2 + plane[1] ** 2 + plane[2] ** 2)
        n_pert = (plane[0] / nn, plane[1] / nn, plane[2] / nn)
        n0 = (plane[0], plane[1], plane[2])
        for z, n in [(depth_rand, n0), (depth, n_rand), (depth_rand, n_rand), (depth, n_pert), (depth, n_pert), (depth_pert, n0)]:
            pl = (n[0], n[1], n[2], distance_to_origin(cams[0], x, y, z, n))
            plane, cost, fragile = consider(pl, plane, cost, fragile)
    return dict(view_weight=vw, selected=selected, plane=tuple(plane), fragile=fragile, wn=wn)


def gen_neighbours(cam, planes_world, weak, nearest_strong, edge, label, label_boundary, complex_val, W, H, x, y, u_limit, search, ransac,
                   depth_min, depth_max, rotate_time, ransac_threshold, use_limit, use_edge, use_label, STRONG=1):
    """GenNeighbours (APD.cu:3330-3711) for one WEAK pixel: the directional search (8 base directions x rotate_time, radii
    2, 4, 8, ... with four randomly shifted tries each, WEAK candidates mapped to their nearest STRONG pixel, duplicate /
    angle / edge-line tests), the label extension along 16 rays, the RANSAC plane through the candidates (inlier count,
    "strong plane" rule, ties by the pixel's own distance) and the ranking of the candidates by their distance to it.
    planes_world: (world normal, depth) per pixel — the pass' input state.  search(k) / ransac(k): the k-th 32-bit numbers
    of the two streams.  Returns (neighbours[1..11], reliable, fragile, tiny): the first `tiny` anchors lie on the fitted plane
    and their mutual order is rounding noise."""
    f32 = np.float32
    fragile = False
    min_margin = 6
    c = y * W + x
    depth_diff = depth_max - depth_min
    K, R = cam["K"], cam["R"]
    angle = 45.0 / rotate_time
    cos_a, sin_a = math.cos(angle * math.pi / 180.0), math.sin(angle * math.pi / 180.0)
    thresh = math.cos((angle / 2.0) * math.pi / 180.0)
    shift_range = max(int(math.tan((angle / 2.0) * math.pi / 180.0) * 20), 1)
    edge_limit = False
    if use_limit:
        edge_limit = True
        if use_edge:
            rp = u_limit - 1.1920929e-07
            if abs(rp - complex_val) < 1e-6:
                fragile = True
            if rp < complex_val:
                edge_limit = False
    pts = [(-1, -1)] * 160
    valid = [False] * 160
    size = 0
    k = 0

    def to_short(v):
        nonlocal fragile
        if 0 < abs(v - round(v)) < 2e-4:
            fragile = True
        return int(v)               # float -> short: truncation toward zero

    def map_strong(q):
        if q[0] < min_margin or q[1] < min_margin or q[0] >= W - min_margin or q[1] >= H - min_margin:
            return None
        if weak[q[0] + q[1] * W] != STRONG:
            q = (int(nearest_strong[q[0] + q[1] * W][0]), int(nearest_strong[q[0] + q[1] * W][1]))
            if q[0] == -1 or q[1] == -1:
                return None
        return q

    odi = -1
    for odx in (-1, 0, 1):
        for ody in (-1, 0, 1):
            if odx == 0 and ody == 0:
                continue
            n = math.hypot(odx, ody)
            od = (odx / n, ody / n)
            odi += 1
            for rot in range(rotate_time):
                di = odi * 4 + rot
                radius = 2
                while radius <= 4096:
                    tx, ty = x + od[0] * radius, y + od[1] * radius
                    if 0 < min(abs(tx), abs(ty), abs(tx - W), abs(ty - H)) < 1e-4:
                        fragile = True
                    if tx < 0 or ty < 0 or tx >= W or ty >= H:
                        break
                    for _ in range(4):
                        sgx = 1 if search(k) % 2 == 0 else 0xFFFFFFFF
                        xs = ((sgx * search(k + 1)) & 0xFFFFFFFF) % shift_range       # unsigned arithmetic: (+-1 * curand) % range (APD.cu:3404)
                        sgy = 1 if search(k + 2) % 2 == 0 else 0xFFFFFFFF
                        ys = ((sgy * search(k + 3)) & 0xFFFFFFFF) % shift_range
                        k += 4
                        d = (od[0] * 20 + xs, od[1] * 20 + ys)
                        dn = math.hypot(d[0], d[1])
                        q = map_strong((to_short(x + d[0] / dn * radius), to_short(y + d[1] / dn * radius)))
                        if q is None:
                            continue
                        if any(pts[j] == q for j in range(di)):
                            continue
                        td = (q[0] - x, q[1] - y)
                        tn = math.hypot(td[0], td[1])
                        cosang = (td[0] * od[0] + td[1] * od[1]) / tn if tn > 0 else float("nan")
                        if abs(cosang - thresh) < 1e-5:
                            fragile = True
                        if cosang > thresh and (not edge_limit or not bresenham_hits_edge(edge, W, H, (x, y), q)):
                            pts[di], valid[di] = q, True
                            size += 1
                            break
                    if valid[di]:
                        break
                    radius = min(radius * 2, radius + 25)
                rd = (od[0] * cos_a - od[1] * sin_a, od[0] * sin_a + od[1] * cos_a)
                rn = math.hypot(rd[0], rd[1])
                od = (rd[0] / rn, rd[1] / rn)
    ext = 31
    if use_label and label[c] > 0:
        dirs = [(0, -1), (0, 1), (-1, 0), (1, 0), (-1, -1), (1, 1), (-1, 1), (1, -1), (1, 0), (0, 1), (0, 1), (-1, 0), (-1, 0), (0, -1), (0, -1), (1, 0)]   # `const int dir[][2]` with 0.5 entries: truncated to 0
        bd, ds = [0.0] * 16, [0] * 16
        for i in range(8):
            b = label_boundary[i]
            dist = 0.0
            if b[0] != -1 and b[1] != -1:
                dist = float(f32(math.sqrt(float((x - b[0]) ** 2 + (y - b[1]) ** 2))))     # `float dist = std::sqrt(std::pow(..) + ..)`: rounded to binary32 ...
                if i >= 4:
                    dist = float(f32(dist / math.sqrt(2.0)))                                # ... and again after `dist /= std::sqrt(2.0)` (48 on a diagonal becomes 47.999996)
            bd[i] = dist
            if i % 2 == 1:
                opp = bd[i - 1]
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = np.float32(4 * rotate_time) * np.float32(bd[i]) / (np.float32(bd[i]) + np.float32(opp))
                ti = int(t) if np.isfinite(t) else -2147483648       # (int)NaN: what the conversion gives
                step = min(1, max(4 * rotate_time - 1, ti))          # MIN(1, MAX(...)): 1 (APD.cu:3478)
                ds[i - 1], ds[i] = 4 * rotate_time - step, step
        for a, (p_, q_) in zip(range(8, 16), ((3, 5), (1, 5), (1, 6), (2, 6), (2, 4), (4, 0), (7, 0), (7, 3))):
            ds[a] = (ds[p_] + ds[q_]) // 2
            bd[a] = float(f32((f32(bd[p_]) + f32(bd[q_])) / f32(2)))
        for i in range(16):
            gap = ds[i] + 1
            q0 = bd[i] / gap
            if 0 < abs(q0 - round(q0)) < 1e-4:
                fragile = True
            step_len = max(1, int(math.floor(1.0 * bd[i] / gap)))
            for step in range(1, ds[i] + 1):
                q = map_strong((x + step * step_len * dirs[i][0], y + step * step_len * dirs[i][1]))
                if q is None:
                    continue
                if any(pts[j] == q for j in range(ext + 1)):
                    continue
                ext += 1
                pts[ext], valid[ext] = q, True
                size += 1
    if size <= 3:
        return None, 0, fragile, 0
    vp = [pts[i] for i in range(160) if valid[i]]
    n = len(vp)

    def cam_point(px_, py_, z):
        return (z * (px_ - K[2]) / K[0], z * (py_ - K[5]) / K[4], z)

    centre_z = float(planes_world[c][3])
    P3 = [cam_point(q[0], q[1], float(planes_world[q[0] + q[1] * W][3])) for q in vp]
    N = []
    for q in vp:
        pw = planes_world[q[0] + q[1] * W].astype(np.float64)
        N.append(tuple(R[3 * r] * pw[0] + R[3 * r + 1] * pw[1] + R[3 * r + 2] * pw[2] for r in range(3)))   # TransformNormal2RefCam
    iteration, max_iter = 300, 200
    max_count = 3
    min_cost = None
    best = None
    has_strong = False
    cache = {}
    k = 0
    fx0, fy0 = (x - K[2]) / K[0], (y - K[5]) / K[4]
    while iteration > 0 and max_iter > 0:
        max_iter -= 1
        ia, ib, ic = ransac(k) % n, ransac(k + 1) % n, ransac(k + 2) % n
        k += 3
        if ia == ib or ib == ic or ia == ic:
            continue
        if not _point_in_triangle(vp[ia], vp[ib], vp[ic], (x, y)):
            continue
        if edge_limit:
            for u, v in ((ia, ib), (ib, ic), (ic, ia)):
                if (u, v) not in cache:
                    cache[(u, v)] = cache[(v, u)] = bresenham_hits_edge(edge, W, H, vp[u], vp[v])
            if cache[(ia, ib)] or cache[(ib, ic)] or cache[(ic, ia)]:
                continue
        AN = N[ia]                                     # AN, BN, CN are all the normal of point a (APD.cu:3594-3596)
        dd = AN[0] * AN[0] + AN[1] * AN[1] + AN[2] * AN[2]
        if abs(dd - 0.9) < 1e-5:
            fragile = True
        if dd < 0.9:
            continue
        A, B, C = P3[ia], P3[ib], P3[ic]
        ac = (A[0] - C[0], A[1] - C[1], A[2] - C[2]); bc = (B[0] - C[0], B[1] - C[1], B[2] - C[2])
        cv = (ac[1] * bc[2] - bc[1] * ac[2], -(ac[0] * bc[2] - bc[0] * ac[2]), ac[0] * bc[1] - bc[0] * ac[1])
        nn = math.sqrt(cv[0] ** 2 + cv[1] ** 2 + cv[2] ** 2)
        if nn == 0 or nn != nn:
            continue
        iteration -= 1
        cv = (cv[0] / nn, cv[1] / nn, cv[2] / nn)
        w = -(cv[0] * A[0] + cv[1] * A[1] + cv[2] * A[2])
        strong_plane = True
        if use_label and label[c] > 0:
            da = abs(AN[0] * cv[0] + AN[1] * cv[1] + AN[2] * cv[2])
            if abs(da - 0.9) < 1e-5:
                fragile = True
            if da < 0.9:
                strong_plane = False
        if has_strong and not strong_plane:
            continue
        cnt = 0
        for si in range(n):
            fx, fy = (vp[si][0] - K[2]) / K[0], (vp[si][1] - K[5]) / K[4]
            dist = abs(-w / (cv[0] * fx + cv[1] * fy + cv[2]) - P3[si][2])
            if abs(dist / depth_diff - ransac_threshold) < 1e-6:
                fragile = True
            if dist / depth_diff < ransac_threshold:
                cnt += 1
        if cnt < 6:
            continue
        cd = abs(-w / (cv[0] * fx0 + cv[1] * fy0 + cv[2]) - centre_z)
        if cnt > max_count or (not has_strong and strong_plane):
            if not has_strong and strong_plane:
                has_strong = True
            best, max_count, min_cost = (cv[0], cv[1], cv[2], w), cnt, cd
        elif cnt == max_count:
            if abs(cd - min_cost) < 1e-6 * max(1.0, min_cost):
                fragile = True
            if cd < min_cost:
                best, min_cost = (cv[0], cv[1], cv[2], w), cd
    if best is None:
        return None, 0, fragile, 0
    wts = []
    out_pts = list(vp)
    for i in range(n):
        fx, fy = (vp[i][0] - K[2]) / K[0], (vp[i][1] - K[5]) / K[4]
        dist = abs(-best[3] / (best[0] * fx + best[1] * fy + best[2]) - P3[i][2])
        if abs(dist / depth_diff - ransac_threshold) < 1e-6:
            fragile = True
        if dist / depth_diff >= ransac_threshold:
            out_pts[i] = (-1, -1)
            wts.append(float("inf"))
        else:
            wts.append(dist)
    order = sorted(range(n), key=lambda i: wts[i])          # insertion sort on strict `<`: stable
    # the three points the plane goes through have distance ~0 — 1e-7 of rounding noise in binary32, 1e-16 here — and lead the
    # list in an order only that noise decides: `tiny` = how many entries form that group (compare it as a set)
    tiny = sum(1 for i in order[:11] if wts[i] < 1e-5)
    for a in range(tiny, min(n - 1, 11)):
        wa, wb = wts[order[a]], wts[order[a + 1]]
        if wa != float("inf") and abs(wa - wb) < 1e-6:
            fragile = True
    if tiny < min(n, 11) and wts[order[tiny]] < 2e-5:
        fragile = True
    ranked = [out_pts[i] for i in order] + [(-1, -1)] * 11
    return ranked[:11], 1, fragile, tiny


def random_initialization(images, cams, x, y, plane, selected, first_init, depth_min, depth_max, top_k, u_depth, u_normal, radius=5, increment=2):
    """RandomInitialization (APD.cu:1273-1309) for one pixel.  FIRST_INIT: a plane whose .w lies outside the depth range is
    replaced by a random hypothesis (GenerateRandomPlaneHypothesis_YZL), one inside is KEPT AS IT IS (a prior: world normal and
    depth-as-offset, the source's behaviour); cost = mean of the top_k smallest NCCs, selected views = those not above the
    k-th (ComputeMultiViewInitialCostandSelectedViews).  Otherwise: (world normal, depth) -> camera-frame plane, cost = mean
    over the selected views whose NCC is below 2, the others dropped with unSetBit — which clears bits 0..n (APD.cu:186-189).
    Returns (plane, cost, selected, fragile)."""
    S = len(cams) - 1
    fragile = False
    if first_init:
        pl = tuple(float(t) for t in plane)
        if plane[3] > depth_max or plane[3] < depth_min:
            z = u_depth * (depth_max - depth_min) + depth_min
            n, fragile = random_normal_yzl(cams, x, y, z, selected, u_normal)
            pl = (n[0], n[1], n[2], distance_to_origin(cams[0], x, y, z, n))
        costs = [ncc_old(images, cams, x, y, v, pl, radius, increment) for v in range(1, S + 1)]
        valid = sum(c < 2.0 for c in costs)
        k = min(valid, top_k)
        if k <= 0:
            return pl, 2.0, 0, fragile
        srt = sorted(costs)
        thr = srt[k - 1]
        if any(c != thr and abs(c - thr) < 2e-4 for c in costs):
            fragile = True
        sel = sum(1 << i for i in range(S) if costs[i] <= thr)
        return pl, sum(srt[:k]) / k, sel, fragile
    R = cams[0]["R"]
    n = tuple(R[3 * r] * plane[0] + R[3 * r + 1] * plane[1] + R[3 * r + 2] * plane[2] for r in range(3))
    pl = (n[0], n[1], n[2], distance_to_origin(cams[0], x, y, float(plane[3]), n))
    sel = int(selected)
    total, count = 0.0, 0
    for i in range(1, S + 1):
        if (sel >> (i - 1)) & 1:
            c = ncc_old(images, cams, x, y, i, pl, radius, increment)
            if c < 2.0:
                count += 1
                total += c
            else:
                sel &= (0xFFFFFFFE << (i - 1)) & 0xFFFFFFFF
    return pl, (total / count if count else 2.0), sel, fragile
