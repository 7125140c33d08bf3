"""Renders longer than the headline frame, timed around DeviceRenderer.trace (best of two; kernel_ms = rttnw_stats): 800x800 spp 5000\n(f64, f32), one rank's share of 1600x1600 spp 10000, 1600x1600 spp 1250 — with a hash of the packed image (RTTNW_CHUNK_SUM_BUDGET=<bytes>\nforces more launches: the hashes must not change).  python profiles/long_render.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np, hashlib
from rttnw_amd import abi, library, render, scene as S
gpu, scenes = library.product(), library.scenes()
torch.cuda.set_device(0)
sc, setup = S.build(gpu, scenes, "final_scene", S.load_earth(), 0)
def run(label, size, spp, prec, rank=0, world=1, reps=2):
    cam, p = S.params_for(setup, size, size, spp, precision=prec, seed=1, tile_rank=rank, tile_world=world)
    r = render.DeviceRenderer(sc, cam, p)
    best = 1e30
    for k in range(reps):
        st = abi.Stats()
        torch.cuda.synchronize(); t0 = time.time()
        r.trace(st); torch.cuda.synchronize()
        best = min(best, (time.time() - t0) * 1e3)
    h = hashlib.sha1(r.packed.cpu().numpy().tobytes()).hexdigest()[:12]
    n = size * size * spp / world
    print("%-34s wall %.1f ms kernel_ms %.1f  %.1f Msamples/s  image %s" % (label, best, st.kernel_ms, n / best / 1e3, h), flush=True)
    del r
run("800x800 spp 1000 f64", 800, 1000, abi.F64)
run("800x800 spp 5000 f64", 800, 5000, abi.F64)
run("800x800 spp 5000 f32", 800, 5000, abi.F32)
run("1600x1600 spp 10000 f64 rank 3/8", 1600, 10000, abi.F64, 3, 8, reps=1)
run("1600x1600 spp 1250 f64 1 rank", 1600, 1250, abi.F64, reps=1)
