"""How evenly does the 8x8-tile interleaving split the frame?  ("final_scene cost per pixel is very uneven".)

No 8-GPU node is available to this build, so this measures on ONE MI355X what bounds the weak-scaling efficiency of
`bench.py --gpus N`: every rank's share of configs[3]'s frame (final_scene 1600x1600, spp = 1250 x N as the bench renders it)
is traced on the same GPU, one after the other; the job ends when the slowest rank does, so
    partition efficiency = mean(rank time) / max(rank time),
and the gather it ends with moves pixels_per_rank x 32 B per rank (f64 RGBA) to rank 0.

    python profiles/partition_balance.py [--spp-per-gpu 1250] [--precision f64]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rttnw_amd import abi, library, render, tiles, scene as S

ap = argparse.ArgumentParser()
ap.add_argument("--spp-per-gpu", type=int, default=1250)
ap.add_argument("--precision", default="f64")
ap.add_argument("--size", type=int, default=1600)
args = ap.parse_args()
prec = abi.F64 if args.precision == "f64" else abi.F32
gpu, scenes = library.product(), library.scenes()
torch.cuda.set_device(0)
sc, setup = S.build(gpu, scenes, "final_scene", S.load_earth(), 0)
print("| ranks | spp | samples per rank | rank kernel ms (min / mean / max) | mean / max | packed bytes per rank |")
print("|---|---|---|---|---|---|")
for world in (1, 2, 4, 8):
    spp = args.spp_per_gpu * world
    ms = []
    for rank in range(world):
        cam, p = S.params_for(setup, args.size, args.size, spp, precision=prec, seed=1, tile_rank=rank, tile_world=world)
        r = render.DeviceRenderer(sc, cam, p)
        if world == 1:  # warm-up: first-use uploads and allocations
            r.trace(); torch.cuda.synchronize()
        st = abi.Stats()
        r.trace(st)  # with stats the call synchronises: rttnw_stats.kernel_ms = HIP events around the trace launches
        torch.cuda.synchronize()
        ms.append(float(st.kernel_ms))
        bytes_rank = r.packed.numel() * r.packed.element_size()
        del r
    print("| %d | %d | %.3g | %.1f / %.1f / %.1f | %.4f | %d |" % (world, spp, args.size * args.size * spp / world, min(ms), sum(ms) / len(ms), max(ms),
                                                                sum(ms) / len(ms) / max(ms), bytes_rank), flush=True)
