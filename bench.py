#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel sample loop on final_scene 800x800 spp=1000 (BASELINE.json).

A "step" is one pass of the hot path over one batch: one full render of the workload (trace kernel(s) +
resolve + tile gather to rank 0 + un-tile/quantise), with the scene already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Arithmetic: the F64 kernels by default — the reference is f64 end to end (src/math/vec3.rs:12); the F32
(throughput) kernels are timed beside them and reported under "f32_kernels".

N = 1 (default): BASELINE.json's metric config, final_scene 800x800 spp=1000.
N > 1: one process per GPU; the 8x8-tile partition of the framebuffer is interleaved over the ranks, every rank
traces its own tiles (no data-path collective), one RCCL gather brings the packed tiles to rank 0.  The workload
is BASELINE configs[3]'s frame, final_scene 1600x1600, at spp = 1250 x N: per-GPU work is fixed (weak scaling:
3200 Msamples per GPU and step) and the N = 8 point IS configs[3] (1600x1600 spp=10000 tile-sharded across 8).
value = samples of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant (trace) kernel against HBM with the counted
algorithmic bytes of SURVEY.md §8(d); `cpu_baseline` times the CPU oracle (a port of the reference: the
Rust reference cannot be built here) on the host cores for a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, width, height, spp per GPU, param)
    "final_scene": ("final_scene", 800, 800, 1000, 0),          # BASELINE.json metric / configs[2] headline
    "cornell_box": ("cornell_box", 800, 800, 1000, 0),          # configs[1]
    "spheres_1m": ("spheres_1m", 1024, 1024, 256, 0),           # configs[4]
    "final_scene_1600": ("final_scene", 1600, 1600, 1250, 0),   # configs[3] when run on 8 GPUs (spp = 1250 x N)
}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
NODE_VISIT_BYTES = 64.0  # accounting price of one visit of a 4-wide (128-byte, four-box) node record: 16 B per child box, as in round 1
PROFILE_ROUND = "r02"
KERNEL_SOURCES = ["rttnw_amd/csrc/render.hip", "rttnw_amd/csrc/rt_core.hpp", "rttnw_amd/csrc/rt_types.hpp", "rttnw_amd/csrc/Makefile"]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha():
    """Identifies the kernel sources a PMC summary was collected with (profiles/collect_pmc.sh records it)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_traffic(scene_name, precision_name):
    """HBM traffic of one launch of the dominant kernel.  PMC counters can only be collected under rocprofv3, so the
    figure comes from the committed summary of `profiles/collect_pmc.sh` for this workload and kernel (FETCH_SIZE x 2
    per the gfx950 correction + WRITE_SIZE, both in KB) — and only while that summary was collected with the kernel
    sources of THIS tree (its `kernel_source_sha` line); otherwise null."""
    pmc = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_%s_%s.txt" % (scene_name, precision_name))
    if not os.path.exists(pmc):
        return None, None
    vals, sha = {}, None
    for line in open(pmc):
        f = line.split()
        if len(f) >= 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"):
            vals[f[0]] = float(f[1])
        if len(f) >= 2 and f[0] == "kernel_source_sha":
            sha = f[1]
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals or sha != kernel_source_sha():
        return None, None
    global _committed_valu
    if "SQ_INSTS_VALU" in vals and vals.get("SQ_ACTIVE_INST_VALU"):
        _committed_valu[(scene_name, precision_name)] = {
            "wave_instructions_per_launch": vals["SQ_INSTS_VALU"],
            "lane_utilisation": round(vals.get("SQ_THREAD_CYCLES_VALU", 0.0) / (vals["SQ_ACTIVE_INST_VALU"] * 64.0), 3)}
    return round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0), "profiles/%s/%s" % (PROFILE_ROUND, os.path.basename(pmc))


_committed_valu = {}   # (scene, precision) -> VALU counters of the same committed summary (filled by committed_traffic)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: final_scene (800x800 spp=1000) on one GPU, final_scene_1600 (spp = 1250 x N) on N > 1")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (per GPU)")
    ap.add_argument("--size", type=int, default=0, help="override width = height")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--counter-spp", type=int, default=8)
    ap.add_argument("--counter-level", type=int, default=1, help="collect_counters of the untimed counting pass (2, 3: more RTTNW_DEBUG_SCHED statistics)")
    ap.add_argument("--no-other", action="store_true", help="skip the timing of the other precision's kernels")
    ap.add_argument("--spp-chunk", type=int, default=0, help="samples per work item (0 = the library's tapered schedule)")
    ap.add_argument("--bvh", default="sah", choices=["sah", "lbvh"], help="BVH builder at commit: host binned SAH (default) or device LBVH")
    ap.add_argument("--share", default=None, metavar="R/W",
                    help="trace ONE rank's share (rank R of a W-way partition) of the N = W workload on this GPU, no gather: "
                         "what each GPU of a W-GPU run does (not a bench line for the driver)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from rttnw_amd import abi, library, render
    from rttnw_amd import scene as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    share = None
    if args.share:
        share = tuple(int(x) for x in args.share.split("/"))
        rank, world = share
    elif world != args.gpus:
        log("bench: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    # RTTNW_BENCH_FORCE_DIST=1: take the torch.distributed code path with a single rank too (checks the launch plumbing
    # on a 1-GPU box; the numbers are the same)
    use_dist = (world > 1 or os.environ.get("RTTNW_BENCH_FORCE_DIST") == "1") and share is None
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
    workload = args.workload or ("final_scene" if world == 1 else "final_scene_1600")
    scene_name, W, H, spp1, param = WORKLOADS[workload]
    if args.size:
        W = H = args.size
    if args.spp:
        spp1 = args.spp
    spp = spp1 * world  # weak scaling: fixed per-GPU work
    precision = abi.F32 if args.precision == "f32" else abi.F64
    earth = S.load_earth()

    t0 = time.time()
    sc, setup = S.build(gpu, scenes, scene_name, earth, param, bvh=abi.BVH_DEVICE_LBVH if args.bvh == "lbvh" else None)
    build_s = time.time() - t0
    binfo = sc.build_info()
    info = abi.Stats()
    gpu.scene_info(sc.handle, info)

    def counted(prec):
        """Algorithmic bytes per sample, counted by the counting kernel variant on this rank's tiles (untimed).
        Node records are priced per child box at the rate round 1 used — its 64-byte two-box record counted as ONE 32-B
        accounting record of BASELINE.md, so the 128-byte four-box record the kernels walk now counts as TWO (64 B): half
        the bytes a visit physically reads, in both rounds."""
        cam_c, pc = S.params_for(setup, W, H, args.counter_spp, precision=prec, tile_rank=rank, tile_world=world, seed=1, collect_counters=args.counter_level)
        rc_ = render.DeviceRenderer(sc, cam_c, pc)
        st = abi.Stats()
        rc_.trace(st)
        n = max(1, st.samples)
        per = dict(rays=st.rays / n, nodes=st.nodes_visited / n, prims=st.prims_tested / n, texels=st.texel_fetches / n)
        return NODE_VISIT_BYTES * per["nodes"] + 32.0 * per["prims"] + 4.0 * per["texels"] + 16.0 / spp, per, st.reserved

    def timed(prec, steps, warmup):
        """`steps` timed steps after `warmup` untimed ones: (ms_per_step over the barrier-bracketed region, mean device
        time of the trace kernel(s) per step from the library's HIP events on the launch stream)."""
        cam_t, p = S.params_for(setup, W, H, spp, precision=prec, tile_rank=rank, tile_world=world, seed=1, spp_chunk=args.spp_chunk)
        r = render.DeviceRenderer(sc, cam_t, p)
        for _ in range(warmup):
            r.trace() if share is not None else r.step()
        st = abi.Stats()
        kms = []
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.trace(st)             # rttnw_stats.kernel_ms: hipEvents around the trace kernel launch(es) on their own stream
            kms.append(st.kernel_ms)
            if share is None:
                r.collect()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        ms_per_step = elapsed * 1e3 / max(1, steps)
        kernel_ms = float(np.mean(kms)) if kms else 0.0
        if use_dist:
            t = torch.tensor([ms_per_step, kernel_ms], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms_per_step, kernel_ms = float(t[0]), float(t[1])
        del r
        return ms_per_step, kernel_ms

    samples_total = float(W) * H * spp          # all ranks together
    samples_rank = samples_total / world
    if share is not None:
        samples_total = samples_rank            # --share: this rank's rate

    def roofline(prec, kernel_ms):
        b_alg, per, form = counted(prec)
        achieved = b_alg * samples_rank / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        pname = "f32" if prec == abi.F32 else "f64"
        # (the committed PMC summaries are of the default configuration: whole frame on one GPU, host SAH trees, default schedule)
        default_run = world == 1 and share is None and not args.spp and not args.size and not args.spp_chunk and args.bvh == "sah"
        traffic, traffic_src = committed_traffic(scene_name, pname) if default_run else (None, None)
        valu = _committed_valu.get((scene_name, pname)) if default_run else None
        if valu is not None:   # the secondary bound of the LDS-resident scenes: what the vector ALUs did, from the same PMC summary
            valu = dict(valu, wave_instructions_per_sample=round(valu["wave_instructions_per_launch"] / samples_rank, 1),
                        note="SQ_INSTS_VALU of one launch and SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU): the kernel is bound by "
                             "vector-instruction issue at this lane utilisation, not by memory")
        # the same time priced at SURVEY 8(d)'s letter — ONE 32-B record per node visit whatever a record holds — for comparison
        strict = (b_alg - (NODE_VISIT_BYTES - 32.0) * per["nodes"]) * samples_rank / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        return {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src, "valu": valu,
                "frac_at_32B_per_node_visit": round(strict / HBM_PEAK_GBPS, 5),
                "kernel": "rt::trace_kernel%s<%s,false>" % ("_plain" if form == 0 else "", "float" if prec == abi.F32 else "double"),
                "kernel_ms": round(kernel_ms, 3), "alg_bytes_per_sample": round(b_alg, 2),
                "alg_bytes_per_launch": round(b_alg * samples_rank),
                "per_sample": {k: round(v, 3) for k, v in per.items()},
                "note": "scene is L2/MALL-resident; achieved = counted algorithmic bytes / kernel time; alg bytes = 64 B x node visits "
                        "(a 128-byte 4-wide record priced at two 32-B accounting records: 16 B per child box, the rate round 1 "
                        "applied to its 64-byte 2-box records) + 32 B x primitive tests + 4 B x texels + 16 B / spp"}

    # ---- timed region (the reported precision)
    ms_per_step, kernel_ms = timed(precision, args.steps, args.warmup)
    value = samples_total / (ms_per_step * 1e-3) / 1e6
    roof = roofline(precision, kernel_ms)

    # ---- the same workload through the other precision's kernels, its own multi-step timing
    other = None
    if not args.no_other:
        oprec = abi.F32 if precision == abi.F64 else abi.F64
        osteps = max(1, min(args.steps, 5))
        oms, okms = timed(oprec, osteps, 1)
        other = {"value": round(samples_total / (oms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(oms, 3),
                 "steps": osteps, "warmup": 1, "roofline": roofline(oprec, okms)}

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from oracle import rto
        so, _ = S.build(rto.binding(), scenes, scene_name, earth, min(param, 20000) if scene_name == "spheres_1m" else param)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        camc, pcal = S.params_for(setup, W, H, 1, seed=1)
        tc = time.perf_counter()
        rto.render(so, camc, pcal, n_threads=cores, want_rgba8=False)      # calibration: 1 spp at full size
        rate = W * H / max(1e-6, time.perf_counter() - tc)
        cspp = int(max(1, min(256, round(rate * args.cpu_seconds / (W * H)))))
        camc, pcpu = S.params_for(setup, W, H, cspp, seed=1)
        tc = time.perf_counter()
        rto.render(so, camc, pcpu, n_threads=cores, want_rgba8=False)
        dt = time.perf_counter() - tc
        cpu = {"value": round(W * H * cspp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": "%s %dx%d spp=%d (%.1f s), f64 CPU oracle (reference-shaped: list scan + reference BVH builder)"
                         % (scene_name, W, H, cspp, dt)}

    if rank == 0 or share is not None:
        out = {
            "metric": "Msamples/sec on final_scene 800x800 spp=1000; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": 1 if share is not None else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s %dx%d spp=%d%s" % (scene_name, W, H, spp, " (spp = %d x %d GPUs)" % (spp1, world) if world > 1 else ""),
                       "max_depth": 50, "scene_seed": "0x5eed0001", "render_seed": 1, "quirks": "reference",
                       "partition": ("8x8 tiles interleaved over %d rank(s), RCCL gather to rank 0" % world) if share is None else
                                    ("--share: rank %d of %d only, on one GPU, no gather; value = this rank's rate" % (rank, world)),
                       "scene_nodes": info.n_nodes, "scene_prims": info.n_prims, "scene_bytes_f32": info.scene_bytes,
                       "scene_build_s": round(build_s, 3), "bvh_builder": args.bvh, "bvh_lower_ms": round(binfo.lower_ms, 2),
                       "bvh_device_ms": round(binfo.device_ms, 3), "stack_depth": binfo.stack_depth},
            "roofline": roof,
            "cpu_baseline": cpu,
            ("f32_kernels" if precision == abi.F64 else "f64_kernels"): other,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
