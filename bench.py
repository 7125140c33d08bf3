#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel sample loop on final_scene 800x800 spp=1000 (BASELINE.json).

A "step" is one pass of the hot path over one batch: one full render of the workload (trace kernel +
resolve + tile gather to rank 0 + un-tile/quantise), with the scene already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; the 8x8-tile partition of the framebuffer is interleaved over the ranks, every
rank traces its own tiles (no data-path collective), one RCCL gather brings the packed tiles to rank 0.
Weak scaling: per-GPU work is fixed — spp grows with N (spp = 1000 N on the same 800x800 image), so the
job is N x 640 Msamples.  value = samples of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant (trace) kernel against HBM with the counted
algorithmic bytes of SURVEY.md §8(d); `cpu_baseline` times the CPU oracle (a port of the reference: the
Rust reference cannot be built here) on the host cores for a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (scene, width, height, spp, param)
    "final_scene": ("final_scene", 800, 800, 1000, 0),      # BASELINE.json metric / configs[2] headline
    "cornell_box": ("cornell_box", 800, 800, 1000, 0),      # configs[1]
    "spheres_1m": ("spheres_1m", 1024, 1024, 256, 0),       # configs[4]
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="final_scene", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (per GPU)")
    ap.add_argument("--size", type=int, default=0, help="override width = height")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--counter-spp", type=int, default=8)
    ap.add_argument("--no-f64", action="store_true", help="skip the one-step F64 cross-check line")
    ap.add_argument("--bvh", default="sah", choices=["sah", "lbvh"], help="BVH builder at commit: host binned SAH (default) or device LBVH")
    args = ap.parse_args()

    import numpy as np
    import torch
    import util
    from rttnw_amd import abi, library, render
    from rttnw_amd import scene as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    # RTTNW_BENCH_FORCE_DIST=1: take the torch.distributed code path with a single rank too (checks the launch plumbing
    # on a 1-GPU box; the numbers are the same)
    use_dist = world > 1 or os.environ.get("RTTNW_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        # RCCL prints its version banner on stdout when NCCL_DEBUG=VERSION is set (it is, on the GPU boxes): keep stdout
        # for the ONE JSON line by pointing fd 1 at stderr while the communicator comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # RCCL over xGMI
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    gpu = library.product()
    if gpu.device_count() < 1:
        raise RuntimeError("bench: no HIP device (no CPU fallback in the product path)")
    scenes = library.scenes()
    scene_name, W, H, spp1, param = WORKLOADS[args.workload]
    if args.size:
        W = H = args.size
    if args.spp:
        spp1 = args.spp
    spp = spp1 * world  # weak scaling: fixed per-GPU work
    precision = abi.F32 if args.precision == "f32" else abi.F64
    earth = S.load_earth()

    t0 = time.time()
    sc, setup = util.build(gpu, scenes, scene_name, earth, param, bvh=abi.BVH_DEVICE_LBVH if args.bvh == "lbvh" else None)
    build_s = time.time() - t0
    binfo = sc.build_info()
    cam, p = util.params_for(setup, W, H, spp, precision=precision, tile_rank=rank, tile_world=world, seed=1)
    info = abi.Stats()
    gpu.scene_info(sc.handle, info)

    # ---- counted algorithmic bytes per sample (untimed, counting kernel variant, this rank's tiles)
    pc = util.params_for(setup, W, H, args.counter_spp, precision=precision, tile_rank=rank, tile_world=world,
                         seed=1, collect_counters=1)[1]
    rc_ = render.DeviceRenderer(sc, cam, pc)
    st = abi.Stats()
    rc_.trace(st)
    n = max(1, st.samples)
    per_sample = dict(rays=st.rays / n, nodes=st.nodes_visited / n, prims=st.prims_tested / n, texels=st.texel_fetches / n)
    b_alg = 32.0 * per_sample["nodes"] + 32.0 * per_sample["prims"] + 4.0 * per_sample["texels"] + 16.0 / spp
    del rc_

    # ---- timed region
    r = render.DeviceRenderer(sc, cam, p)
    for _ in range(args.warmup):
        r.step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()          # HIP events on the stream the trace kernel is launched on
        r.trace()
        ev[k][1].record()
        r.collect()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else 0.0
    ms_per_step = elapsed * 1e3 / max(1, args.steps)
    if use_dist:
        t = torch.tensor([ms_per_step, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_per_step, kernel_ms = float(t[0]), float(t[1])

    samples_total = float(W) * H * spp          # all ranks together
    samples_rank = samples_total / world
    value = samples_total / (ms_per_step * 1e-3) / 1e6
    achieved = b_alg * samples_rank / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    # ---- the same workload through the F64 kernels (the reference's arithmetic type), one step, N = 1 only
    f64 = None
    if world == 1 and precision == abi.F32 and not args.no_f64:
        cam64, p64 = util.params_for(setup, W, H, spp, precision=abi.F64, seed=1)
        r64 = render.DeviceRenderer(sc, cam64, p64)
        r64.step()
        torch.cuda.synchronize()
        tq = time.perf_counter()
        r64.step()
        torch.cuda.synchronize()
        dq = time.perf_counter() - tq
        f64 = {"value": round(samples_total / dq / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dq * 1e3, 3), "steps": 1}
        del r64

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import rto
        so, _ = util.build(rto.binding(), scenes, scene_name, earth, min(param, 20000) if scene_name == "spheres_1m" else param)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        camc, pcal = util.params_for(setup, W, H, 1, seed=1)
        tc = time.perf_counter()
        rto.render(so, camc, pcal, n_threads=cores, want_rgba8=False)      # calibration: 1 spp at full size
        rate = W * H / max(1e-6, time.perf_counter() - tc)
        cspp = int(max(1, min(256, round(rate * args.cpu_seconds / (W * H)))))
        camc, pcpu = util.params_for(setup, W, H, cspp, seed=1)
        tc = time.perf_counter()
        rto.render(so, camc, pcpu, n_threads=cores, want_rgba8=False)
        dt = time.perf_counter() - tc
        cpu = {"value": round(W * H * cspp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": "%s %dx%d spp=%d (%.1f s), f64 CPU oracle (reference-shaped: list scan + reference BVH builder)"
                         % (scene_name, W, H, cspp, dt)}

    # HBM traffic of one launch of the dominant kernel: PMC counters can only be collected under rocprofv3, so the
    # figure is read from the committed summary of `profiles/collect_pmc.sh` for this workload and kernel
    # (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, both in KB), or null when there is none.
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "r01", "pmc_%s_%s.txt" % (scene_name, args.precision))
    if world == 1 and not args.spp and not args.size and os.path.exists(pmc):
        vals = {}
        for line in open(pmc):
            f = line.split()
            if len(f) >= 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[f[0]] = float(f[1])
        if len(vals) == 2:
            traffic = round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
            traffic_src = "profiles/r01/" + os.path.basename(pmc)

    if rank == 0:
        out = {
            "metric": "Msamples/sec on final_scene 800x800 spp=1000; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s %dx%d spp=%d%s" % (scene_name, W, H, spp, " (spp = %d x %d GPUs)" % (spp1, world) if world > 1 else ""),
                       "max_depth": 50, "scene_seed": "0x5eed0001", "render_seed": 1, "quirks": "reference",
                       "partition": "8x8 tiles interleaved over %d rank(s), RCCL gather to rank 0" % world,
                       "scene_nodes": info.n_nodes, "scene_prims": info.n_prims, "scene_bytes_f32": info.scene_bytes,
                       "scene_build_s": round(build_s, 3), "bvh_builder": args.bvh, "bvh_lower_ms": round(binfo.lower_ms, 2),
                       "bvh_device_ms": round(binfo.device_ms, 3), "stack_depth": binfo.stack_depth},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "rt::trace_kernel%s<%s,false>" % ("_plain" if st.reserved == 0 else "", "float" if precision == abi.F32 else "double"),
                         "kernel_ms": round(kernel_ms, 3),
                         "alg_bytes_per_sample": round(b_alg, 2),
                         "per_sample": {k: round(v, 3) for k, v in per_sample.items()},
                         "note": "scene is L2/MALL-resident; achieved = counted algorithmic bytes / kernel time"},
            "cpu_baseline": cpu,
            "f64_kernels": f64,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
