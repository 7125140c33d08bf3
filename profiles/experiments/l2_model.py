#!/usr/bin/env python3
"""L2-miss lines per sample of the decoupled kernel on spheres_1m, by stream, for candidate layouts of the node and sphere records — the host
model of tests/hostsim/cache_model.hpp (test infrastructure; nothing here is product code).  Usage:
    python3 profiles/experiments/l2_model.py [--spheres N] [--waves W] [--cache-mb M] [--layouts current,bfs,...]
The tree is the HOST builder's (binned SAH + collapse4: the device SAH tree renders within 0.5 % of it)."""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rttnw_amd import abi, library           # noqa: E402
from rttnw_amd import scene as S             # noqa: E402

RAYS_PER_SAMPLE = 7.03
STREAMS = ["nodes", "spheres", "sphere_mat", "materials", "pool_hot", "pool_cold", "job_sums"]


def hostsim_binding():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hostsim"), "-s"], check=True)
    lib = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"))
    b = abi.Binding(lib, "rttnw_", abi.BUILDER_FUNCS)
    b.add([("builder", C.c_void_p, [])])
    return b


def tree_arrays(hs, sc):
    dims = (C.c_uint32 * 8)()
    hs.lib.hostsim_scene_dims(sc.handle, dims)
    n = dims[0]
    child = np.zeros((n, 4), dtype=np.int32)
    area = np.zeros((n, 4), dtype=np.float32)
    root = C.c_int32()
    hs.lib.hostsim_nodes(sc.handle, child.ctypes.data_as(C.c_void_p), area.ctypes.data_as(C.c_void_p), C.byref(root))
    return child, area, root.value, dims[1]


# ---- layouts: perm[i] = where record i lies -----------------------------------------------------------------------------------------
def order_to_perm(order, n):
    perm = np.full(n, -1, dtype=np.int64)
    perm[np.asarray(order, dtype=np.int64)] = np.arange(len(order))
    rest = np.flatnonzero(perm < 0)
    perm[rest] = np.arange(len(order), len(order) + len(rest))   # records no walk reaches (instance trees of other scenes)
    return perm.astype(np.uint32)


def layout_bfs(child, area, root, by_area=False):
    order, frontier = [], [root]
    while frontier:
        order.extend(frontier)
        nxt = []
        for i in frontier:
            cs = [(c, child[i, c]) for c in range(4) if child[i, c] >= 0]
            if by_area:
                cs.sort(key=lambda t: -area[i, t[0]])
            nxt.extend(ch for _, ch in cs)
        frontier = nxt
    return order


def layout_dfs(child, area, root, by_area=False):
    order, stack = [], [root]
    while stack:
        i = stack.pop()
        order.append(i)
        cs = [(c, child[i, c]) for c in range(4) if child[i, c] >= 0]
        if by_area:
            cs.sort(key=lambda t: -area[i, t[0]])
        stack.extend(ch for _, ch in reversed(cs))
    return order


def subtree_sizes(child, root):
    n = child.shape[0]
    size = np.ones(n, dtype=np.int64)
    order = layout_bfs(child, None, root)
    for i in reversed(order):
        for c in range(4):
            if child[i, c] >= 0:
                size[i] += size[child[i, c]]
    return size


def layout_treelets(child, area, root, treelet, by_area=True):
    """Treelets of up to `treelet` records: a treelet grows from its root by taking, again and again, the frontier record whose box has the
    largest surface area (the one a ray that reached the treelet most probably visits next); its records lie contiguously in the order taken;
    the records left on the frontier root the next treelets, laid out depth-first (a treelet next to its first child treelet)."""
    import heapq
    order, stack = [], [root]
    while stack:
        r = stack.pop()
        taken, heap = [], [(-1e30, r)]
        while heap and len(taken) < treelet:
            _, i = heapq.heappop(heap)
            taken.append(i)
            for c in range(4):
                if child[i, c] >= 0:
                    heapq.heappush(heap, (-float(area[i, c]) if by_area else len(taken), child[i, c]))
        order.extend(taken)
        rest = sorted(heap)                       # largest area first
        stack.extend(i for _, i in reversed(rest))
    return order


def layout_pairs(child, area, root):
    """Lines of two: depth-first, but a record's inner children are emitted as adjacent PAIRS (largest-area two together) right after it when it
    is line-aligned — so that the two most probable next visits share ONE line."""
    order, stack = [], [("n", root)]
    placed = set()
    while stack:
        kind, i = stack.pop()
        if kind == "n":
            if i not in placed:
                order.append(i); placed.add(i)
            cs = sorted([(-area[i, c], child[i, c]) for c in range(4) if child[i, c] >= 0])
            kids = [ch for _, ch in cs]
            # emit the children now, as a block of siblings (pairs of siblings share lines when the block starts on a line)
            if len(order) % 2 == 1 and kids:
                pass
            for ch in kids:
                order.append(ch); placed.add(ch)
            stack.extend(("n", ch) for ch in reversed(kids))
    return order


def sphere_order_from_tree(child, order_nodes, n_spheres):
    """Sphere records in the order the node layout meets their leaves (a leaf = up to four consecutive same-kind records)."""
    seen = np.zeros(n_spheres, dtype=bool)
    out = []
    for i in order_nodes:
        for c in range(4):
            ch = int(child[i, c])
            if ch < 0 and ch != -2 ** 31:
                bits = (~ch) & 0xFFFFFFFF
                kind, count, first = bits >> 28, ((bits >> 26) & 3) + 1, bits & 0x3FFFFFF
                if kind == 0:
                    for k in range(first, first + count):
                        if not seen[k]:
                            seen[k] = True
                            out.append(k)
    rest = np.flatnonzero(~seen)
    return np.concatenate([np.asarray(out, dtype=np.int64), rest])


def unified_layout(child, order_nodes, n_spheres, align_parent=False, unit=32):
    """Node and sphere records in ONE buffer, positions in units of a sphere record (32 B in f64, 16 B in f32): every node record (64 B, on a 64-byte
    boundary) is followed by the sphere records of its sphere leaves.  align_parent: a record that has sphere leaves starts on a 128-byte line (its
    first two — f32: all four — spheres share its line)."""
    nu, lu = 64 // unit, 128 // unit
    n = child.shape[0]
    npos = np.zeros(n, dtype=np.uint32)
    spos = np.full(n_spheres, 0xFFFFFFFF, dtype=np.uint32)
    at = 0
    for i in order_nodes:
        leaves = []
        for c in range(4):
            ch = int(child[i, c])
            if ch < 0 and ch != -2 ** 31:
                bits = (~ch) & 0xFFFFFFFF
                if bits >> 28 == 0:
                    leaves.append((bits & 0x3FFFFFF, ((bits >> 26) & 3) + 1))
        at += (-at) % nu
        if align_parent and leaves and (at % lu):
            at += lu - (at % lu)
        npos[i] = at
        at += nu
        for first, count in leaves:
            for k in range(first, first + count):
                spos[k] = at
                at += 1
    rest = np.flatnonzero(spos == 0xFFFFFFFF)
    spos[rest] = at + np.arange(len(rest), dtype=np.uint32)
    print("    unified buffer: %.1f MB" % ((at + len(rest)) * unit / 1e6))
    return npos, spos


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spheres", type=int, default=1000000)
    ap.add_argument("--waves", type=int, default=416, help="waves sharing the cache: 32 CUs x 13 waves per XCD on the f64 LEAN flavour")
    ap.add_argument("--cache-mb", type=float, default=4.0)
    ap.add_argument("--ways", type=int, default=16)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--measure", type=int, default=4)
    ap.add_argument("--layouts", default="current,dfs_area,bfs,bfs_area,treelet8,treelet16,treelet32")
    ap.add_argument("--variants", default="base", help="comma list of: base, spheres_tree (sphere records in node-layout order), "
                    "mat_by_sphere (material beside the sphere's index, 64 B), precull (one-sphere leaves pre-tested, 5 %% inflated)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"], help="record sizes and the walk's arithmetic (f32: 16-byte spheres, 512 waves per XCD: pass --waves 512)")
    ap.add_argument("--batch-stride", type=int, default=0, help="0: job batches strided over the whole render; k: every k-th batch from --batch-offset (the kernel's order: k = 8 XCDs)")
    ap.add_argument("--batch-offset", type=int, default=0)
    ap.add_argument("--node-steps", type=int, default=5, help="node steps per trip (RT_WAVE_STEPS)")
    ap.add_argument("--retire", type=int, default=8, help="finished rays that end a burst while rays are queued (RT_WAVE_RETIRE)")
    ap.add_argument("--cold-words", type=int, default=5, help="words of slot bookkeeping read at a path's end (5: the kernel's; 2: the job as its index — round 6, measured slower)")
    ap.add_argument("--leaf-threshold", type=int, default=0, help="dynamic leaf steps inside a trip once this many lanes wait at a leaf (0: the kernel's fixed trips)")
    args = ap.parse_args()

    hs = hostsim_binding()
    scenes = library.scenes()
    t0 = time.time()
    sc, setup = S.build(hs, scenes, "spheres_1m", None, args.spheres)   # (the host build has only the host builder)
    child, area, root, n_spheres = tree_arrays(hs, sc)
    n = child.shape[0]
    print("scene: %d spheres, %d four-wide records, built in %.1f s" % (n_spheres, n, time.time() - t0), flush=True)
    cam, p = S.params_for(setup, args.size, args.size, 256, precision=abi.F32 if args.precision == "f32" else abi.F64, seed=1)

    def layout(name):
        if name == "current":
            return list(range(n)), None
        if name == "dfs_area":
            o = layout_dfs(child, area, root, True)
        elif name == "dfs":
            o = layout_dfs(child, area, root, False)
        elif name == "bfs":
            o = layout_bfs(child, area, root, False)
        elif name == "bfs_area":
            o = layout_bfs(child, area, root, True)
        elif name.startswith("treelet"):
            o = layout_treelets(child, area, root, int(name[7:]))
        elif name == "pairs":
            o = layout_pairs(child, area, root)
        else:
            raise SystemExit("unknown layout " + name)
        return o, order_to_perm(o, n)

    print("%-12s %-14s | %s | total rd-miss  wr-back | per ray: nodes sph-tests skipped" % ("layout", "variant", " ".join("%10s" % s for s in STREAMS)))
    for lname in args.layouts.split(","):
        order, perm = layout(lname)
        for variant in args.variants.split(","):
            vs = set(variant.split("+"))
            sperm = None
            sph_b = 16 if args.precision == "f32" else 32
            unified = "unified" in vs
            nperm = perm
            if unified:
                nperm, sperm = unified_layout(child, order, n_spheres, align_parent="aligned" in vs, unit=sph_b)
            elif "spheres_tree" in vs:
                so = sphere_order_from_tree(child, order, n_spheres)
                sperm = order_to_perm(so, n_spheres)
            f32 = args.precision == "f32"
            prm = np.zeros(20, dtype=np.uint32)
            cache_bytes = int(args.cache_mb * (1 << 20))
            prm[:] = [args.waves, args.ways, args.node_steps, args.retire, args.warm, args.measure, cache_bytes & 0xFFFFFFFF, cache_bytes >> 32,
                      sph_b, (32 if f32 else 64) if "mat_by_sphere" in vs else (24 if f32 else 40), 1 if "mat_by_sphere" in vs else 0, 4 if f32 else 8, 1 if "precull" in vs else 0, 5, args.batch_stride, 64, args.batch_offset, 1 if unified else 0, args.leaf_threshold, args.cold_words]
            out = np.zeros(192, dtype=np.uint64)
            t0 = time.time()
            hs.lib.hostsim_cache_model(sc.handle, C.byref(cam), C.byref(p), prm.ctypes.data_as(C.c_void_p),
                                       nperm.ctypes.data_as(C.c_void_p) if nperm is not None else None,
                                       sperm.ctypes.data_as(C.c_void_p) if sperm is not None else None, out.ctypes.data_as(C.c_void_p))
            # normalised per RAY and quoted per sample at the measured rays per sample (profiles/r05: 7.03 on spheres_1m): a model run is short against
            # the ~53 000 paths in flight, so the samples it COMPLETES are biased to the short ones (sky) while the lines per ray are not
            ns = float(out[1]) / RAYS_PER_SAMPLE
            miss = [out[9 + 4 * s] / ns for s in range(len(STREAMS))]
            wb = sum(out[10 + 4 * s] for s in range(len(STREAMS))) / ns
            print("%-12s %-14s | %s | %8.1f %8.1f | %6.1f %6.2f %6.2f   (%.0f s, %d samples)" %
                  (lname, variant, " ".join("%10.2f" % m for m in miss), sum(miss), wb, out[2] / max(1.0, float(out[1])), out[3] / max(1.0, float(out[1])),
                   out[4] / max(1.0, float(out[1])), time.time() - t0, int(out[1])), flush=True)
            ne, nl, le, ll, se, sl = [float(out[128 + k]) for k in range(6)]
            # wave-level cost of the schedule in vector instructions per sample (quantised node step ~170, sphere leaf step ~120, a shade pass ~1500)
            print("    steps per sample: node %.2f executions x %.1f lanes, leaf %.2f x %.1f, shade %.3f x %.1f  -> ~%.0f instructions per sample"
                  % (ne / ns, nl / max(1.0, ne), le / ns, ll / max(1.0, le), se / ns, sl / max(1.0, se), (ne * 170 + le * 120 + se * 1500) / ns), flush=True)
            if lname == "current" and variant == "base":
                print("    node accesses / misses per sample by depth: " +
                      " ".join("%d:%.1f/%.1f" % (d, out[64 + d] / ns, out[96 + d] / ns) for d in range(32) if out[64 + d]), flush=True)


if __name__ == "__main__":
    main()
