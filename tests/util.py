"""Helpers shared by the tests."""
import ctypes as C

import numpy as np

from rttnw_amd import abi
from rttnw_amd import scene as S


build, params_for = S.build, S.params_for  # (live in the package: bench.py and smoke() use them too)


def hostsim_render(hostsim, sc, cam, p, n_threads=0):
    lin = np.zeros((p.height, p.width, 3))
    st = abi.Stats()
    rc = hostsim.lib.hostsim_render(sc.handle, C.byref(cam), C.byref(p), lin.ctypes.data, C.byref(st), n_threads)
    assert rc == 0
    return lin, st


def block_means(rgb, n):
    h, w, _ = rgb.shape
    bh, bw = h // n, w // n
    return np.array([[[rgb[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw, ch].mean() for c in range(n)]
                      for r in range(n)] for ch in range(3)])


# ---- per-bounce path dumps: product (device probe kernel, or the host build of the same core) vs the oracle
PROBE_STRIDE = 20  # include/rttnw_hip.h rttnw_debug_probe_path


def product_probe(fn, binding, sc, cam, p, px, row, sample, max_out=64):
    """fn = gpu.debug_probe_path or hostsim.lib.hostsim_probe_path -> array [n_hits, 20]."""
    out = np.zeros(max_out * PROBE_STRIDE + 4, dtype=np.float64)
    n = fn(sc.handle, C.byref(cam), C.byref(p), px, row, sample, out.ctypes.data, max_out)
    abi.check(n, binding, "probe_path")
    return out[:n * PROBE_STRIDE].reshape(n, PROBE_STRIDE)


def compare_paths(probe, oracle_probe, pairs, tol=1e-9, growth=1.0):
    """Every bounce of every (px, row, sample): t, p, normal, front_face equal within `tol` (relative to the magnitude of
    the coordinate) AT EVERY DEPTH.  `growth` > 1 — only for the contracted build on the scenes whose small spheres amplify a last-place
    difference (the world-space test of a transformed group's spheres, a fused multiply-add: a different ray after the bounce, growing by
    ~(1 + distance / radius) per bounce, x10 in final_scene's cluster) — loosens the bound to tol x growth^(k - 2) at bounce k > 2, capped
    at 1e-4; RTTNW_F64_STRICT probes are held to 1e-12 flat (tests/test_gpu_parity.py).  The DECISIONS — front face, material,
    scattered or absorbed, bounce count — are exact at every depth; the same material at every hit (the oracle reports graph ids, the product flat indices: the
    mapping must be one-to-one and order preserving), (u, v) equal wherever the product computes them (it skips them
    when no texture reads them).  Returns (paths, bounces compared, material map)."""
    mat_map, bounces = {}, 0
    for (px, row, s) in pairs:
        a = probe(px, row, s)
        b = oracle_probe(px, row, s)
        assert len(a) == len(b), ("bounce count", px, row, s, len(a), len(b))
        for k in range(len(a)):
            scale = max(1.0, np.abs(b[k, 0:4]).max())
            tol_k = min(max(tol, 1e-4), tol * growth ** max(0, k - 2))
            assert np.abs(a[k, 0:7] - b[k, 0:7]).max() <= tol_k * scale, ("t/p/normal", px, row, s, k, a[k, 0:7], b[k, 0:7])
            assert a[k, 10] == b[k, 10], ("front_face", px, row, s, k)
            if a[k, 8] != 0.0 or a[k, 9] != 0.0:
                assert abs(a[k, 8] - b[k, 8]) <= tol_k and abs(a[k, 9] - b[k, 9]) <= tol_k, ("uv", px, row, s, k)
            assert mat_map.setdefault(int(b[k, 7]), int(a[k, 7])) == int(a[k, 7]), ("material", px, row, s, k)
            assert (a[k, 19] >= 0.0) == (b[k, 11] == 1.0), ("scattered", px, row, s, k)
            bounces += 1
    ids = sorted(mat_map)
    flat = [mat_map[i] for i in ids]
    assert flat == sorted(flat) and len(set(flat)) == len(flat), mat_map
    return len(pairs), bounces, mat_map


# ---- node records (include/rttnw_hip.h rttnw_debug_scene_nodes / rttnw_debug_scene_nodes4)
NODE2 = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3), ("child", "<i4", 2), ("pad", "<i4", 2)])
NODE4 = np.dtype([("lo", "<f4", (3, 4)), ("hi", "<f4", (3, 4)), ("child", "<i4", 4), ("pad", "<i4", 4)])
assert NODE2.itemsize == 64 and NODE4.itemsize == 128
CHILD_EMPTY = -2**31


def nodes_of(binding, sc, wide=False):
    fn = binding.debug_scene_nodes4 if wide else binding.debug_scene_nodes
    n = fn(sc.handle, None, 0, None)
    buf = np.zeros(n, dtype=NODE4 if wide else NODE2)
    root = C.c_int32()
    assert fn(sc.handle, buf.ctypes.data, n, C.byref(root)) == n
    return buf, root.value


def leaves_of_binary(nodes, root):
    out, stack = [], [root]
    while stack:
        nd = nodes[stack.pop()]
        for ch in nd["child"]:
            if ch >= 0:
                stack.append(int(ch))
            elif ch != CHILD_EMPTY:
                out.append(int(ch))
    return sorted(out)


def check_wide_tree(nodes4, root):
    """Walk a 4-wide tree: every record reached once, a child's boxes inside the slot box its parent holds for it, unused
    slots empty.  Returns (sorted leaf codes, stack entries a walk can have pending = what the lowering must bound)."""
    leaves, seen = [], set()

    def visit(i, plo, phi):
        assert i not in seen
        seen.add(i)
        nd = nodes4[i]
        k, deepest = 0, 0
        for c in range(4):
            ch, lo, hi = int(nd["child"][c]), nd["lo"][:, c], nd["hi"][:, c]
            if ch == CHILD_EMPTY:
                assert (lo > hi).all()
                continue
            k += 1
            assert (lo <= hi).all() and (lo >= plo).all() and (hi <= phi).all()
            if ch >= 0:
                deepest = max(deepest, visit(ch, lo, hi))
            else:
                leaves.append(ch)
        return max(k - 1, 0) + deepest

    need = visit(root, np.full(3, -np.inf, np.float32), np.full(3, np.inf, np.float32))
    return sorted(leaves), need, seen
