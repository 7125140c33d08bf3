"""Commit the 10^6-sphere scene of BASELINE config 5 through the two device builders (twice each: the second is warm) — the
program `rocprofv3 --kernel-trace --stats` is run on for profiles/r03/kernel_stats_device_builders.csv:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 profiles/device_build_trace.py
RTTNW_DEBUG_LOWER=1 prints the phases of every commit on stderr."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from rttnw_amd import abi, library, scene as S
gpu, scenes = library.product(), library.scenes()
torch.cuda.set_device(0)
torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
for label, bvh in [("dsah", abi.BVH_DEVICE_SAH), ("dsah", abi.BVH_DEVICE_SAH), ("lbvh", abi.BVH_DEVICE_LBVH), ("lbvh", abi.BVH_DEVICE_LBVH)]:
    t0 = time.time()
    sc = S.Scene(gpu, 0x5EED0001, scenes_binding=scenes)
    sc.set_bvh_builder(bvh)
    sc.build_named("spheres_1m", param=0)
    bi = sc.build_info()
    print("%s: describe + commit %.1f ms, commit (lower_ms) %.1f ms, device %.2f ms, %d 4-wide records" % (label, (time.time() - t0) * 1e3, bi.lower_ms, bi.device_ms, bi.n_nodes), flush=True)
