#!/usr/bin/env python3
"""bench.py — Msamples/s of the per-pixel sample loop on final_scene 800x800 spp=1000 (BASELINE.json).

A "step" is one pass of the hot path over one batch: one full render of the workload (trace kernel(s) +
resolve + tile gather to rank 0 + un-tile/quantise), with the scene already resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Arithmetic: the F64 kernels by default — the reference is f64 end to end (src/math/vec3.rs:12); the F32
(throughput) kernels are timed beside them and reported under "f32_kernels".

N = 1 (default): BASELINE.json's metric config, final_scene 800x800 spp=1000, in the top-level fields; the other two
single-GPU configs of BASELINE.json — cornell_box 800x800 spp=1000 (configs[1]) and spheres_1m 1024x1024 spp=256
(configs[4]) — are timed in the same run with a few steps each and reported as the sub-records "cornell_box" and
"spheres_1m" (own steps, ms_per_step, kernel_ms, roofline; --no-sub skips them).
N > 1: one process per GPU; the 8x8-tile partition of the framebuffer is interleaved over the ranks, every rank
traces its own tiles (no data-path collective), one RCCL gather brings the packed tiles to rank 0.  The workload
is BASELINE configs[3]'s frame, final_scene 1600x1600, at spp = 1250 x N: per-GPU work is fixed (weak scaling:
3200 Msamples per GPU and step) and the N = 8 point IS configs[3] (1600x1600 spp=10000 tile-sharded across 8).
value = samples of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line of at most 6 KB on stdout: the contract's top-level fields, a FLAT `roofline`, `cpu_baseline`, the
headline workload through the other two builds as flat scalars (`strict_value` / `strict_ms_per_step`: RTTNW_F64_STRICT, the build whose
pixels ARE the reference's; `f32_value` / `f32_ms_per_step`) and, under `sub`, one flat record per sub-workload and build.  Everything
else (notes, instruction classes, per-sample counts, the other readings of the byte accounting) goes to `gpurun_out/bench_detail.json`
(RTTNW_BENCH_DETAIL=<path> to move it) and, as one line, to stderr.

`roofline` says what bounds the dominant (trace) kernel:
  * scenes whose node records live in LDS (final_scene, cornell_box): "bound": "valu" — achieved = full-wave VALU
    instructions per second (SQ_INSTS_VALU x lane utilisation / kernel time, from the committed rocprofv3 PMC summary of
    this exact kernel source) against the issue peak of MI355X_MICROARCH.md (1024 SIMDs x 2.4 GHz; 2 cycles per wave64 f32
    instruction, 4 per f64).  HBM is not what bounds them: counter traffic is ~1 % of peak.
  * spheres_1m (node records in HBM / Infinity Cache): "bound": "hbm" — achieved = ALGORITHMIC bytes / kernel time with
    SURVEY 8(d)'s accounting to the letter: 32 B per node visit + 32 B per primitive test + 4 B per texel + 16 B / spp.
  Every record carries `hbm_algorithmic` (the 8(d) reading at 32 B per visit, and the same time priced at 64 B — 16 B per
  child box — and at the 128 B a 4-wide record physically is) and `traffic` (HBM bytes per launch from the PMC counters,
  FETCH_SIZE x 2 + WRITE_SIZE, or null when no summary of this kernel source is committed).
`cpu_baseline` times the CPU oracle (a port of the reference: the Rust reference cannot be built here) on the host
cores for a bounded sample of the headline workload.
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
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
N_SIMDS = 1024           # 256 CUs x 4 SIMDs (same table)
# What a wave64 vector instruction costs a SIMD in issue CYCLES, by the classes the SQ_INSTS_VALU_* counters distinguish: MEASURED on an
# MI355X (profiles/valu_cost.hip under rocprofv3, profiles/valu_cost_cycles.sh -> profiles/r04/valu_cost_cycles.txt: one long launch per
# kind at 8 waves per SIMD, GRBM_GUI_ACTIVE / instructions a SIMD issued, 2.38-2.40 GHz): v_fma_f32 2.36 and plain logic 2.28; f64
# fma / add / mul 4.16 — and so do v_mul_lo / hi_u32, v_cndmask_b32, v_cmp_*, v_cvt_*, v_bcnt, v_lshrrev_b64, v_max3_f32 (4.16-4.5);
# v_mad_u64_u32 4.6; v_rcp_f32 8.2; v_rcp_f64 / v_sqrt_f64 16.2.  INT32 holds logic and adds (2.3) as well as multiplies, selects and popcounts
# (4.2): priced at their mean; the unclassified rest (moves 2.3; compares, selects, min / max 4.2) likewise.
CLOCK_GHZ = 2.4          # max shader clock (MI355X_MICROARCH.md); the profiled launches ran at 2.38
VALU_COST_CYCLES = {"SQ_INSTS_VALU_FMA_F32": 2.36, "SQ_INSTS_VALU_MUL_F32": 2.36, "SQ_INSTS_VALU_ADD_F32": 2.36, "SQ_INSTS_VALU_TRANS_F32": 8.2,
                    "SQ_INSTS_VALU_FMA_F64": 4.16, "SQ_INSTS_VALU_MUL_F64": 4.16, "SQ_INSTS_VALU_ADD_F64": 4.16, "SQ_INSTS_VALU_TRANS_F64": 16.2,
                    "SQ_INSTS_VALU_INT32": 3.3, "SQ_INSTS_VALU_INT64": 4.4, "SQ_INSTS_VALU_CVT": 4.3}
VALU_COST_OTHER_CYCLES = 3.3
NODE_VISIT_BYTES = 32.0  # ONE 32-B accounting record per node visit, whatever a record physically holds
PROFILE_ROUND = "r06"
# (what a launch's counters depend on: the kernels, their launch code and flags — and, since round 6, what lays the records out for them: the device
# collapse's numbering and the upload)
KERNEL_SOURCES = ["rttnw_amd/csrc/trace_kernels.hpp", "rttnw_amd/csrc/render_tiles.hpp", "rttnw_amd/csrc/rt_core.hpp", "rttnw_amd/csrc/rt_types.hpp",
                  "rttnw_amd/csrc/bvh_quant.hpp", "rttnw_amd/csrc/Makefile", "rttnw_amd/csrc/bvh_build.hip", "rttnw_amd/csrc/render_common.hpp"]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha():
    """Identifies the kernel sources a PMC summary was collected with (profiles/collect_pmc.sh records it)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_pmc(scene_name, precision_name):
    """Counters of one launch of the dominant kernel.  PMC counters can only be collected under rocprofv3, so they come
    from the committed summary of `profiles/collect_pmc.sh` for this workload and kernel — and only while that summary
    was collected with the kernel sources of THIS tree (its `kernel_source_sha` line); otherwise None."""
    pmc = os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_%s_%s.txt" % (scene_name, precision_name))
    if not os.path.exists(pmc):
        return None
    vals, sha = {}, None
    for line in open(pmc):
        f = line.split()
        if len(f) >= 2 and f[0] == "kernel_source_sha":
            sha = f[1]
        elif len(f) >= 2 and f[0].isupper():
            try:
                vals[f[0]] = float(f[1])
            except ValueError:
                pass
    if sha != kernel_source_sha():
        return None
    vals["source"] = "profiles/%s/%s" % (PROFILE_ROUND, os.path.basename(pmc))
    return vals


FLAT_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "hbm_alg_frac", "hbm_alg_gbps", "alg_bytes_per_sample", "traffic",
                      "traffic_over_algorithmic", "lane_utilisation", "issue_utilisation", "valu_insts_per_sample", "kernel", "kernel_ms", "pmc")
LINE_LIMIT = 6000   # bytes of the ONE stdout line (the driver keeps an 8 KB tail of stdout; round 5's 25.9 KB line could not be parsed)


def flat_roofline(detail):
    """The scalars of a roofline record, no nesting and no prose (the full record goes to bench_detail.json)."""
    return {k: detail.get(k) for k in FLAT_ROOFLINE_KEYS}


def flat_sub(rec, cpu=None):
    """One flat record per sub-workload and build."""
    r = rec["roofline"]
    out = {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"], "kernel_ms": r["kernel_ms"],
           "hbm_alg_frac": r["hbm_alg_frac"], "traffic_over_algorithmic": r["traffic_over_algorithmic"],
           "lane_utilisation": r["lane_utilisation"], "bound": r["bound"], "frac": r["frac"]}
    if cpu is not None:
        out["cpu_value"] = cpu["value"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: final_scene (800x800 spp=1000) on one GPU, final_scene_1600 (spp = 1250 x N) on N > 1")
    ap.add_argument("--precision", default="f64", choices=["f32", "f64", "f64strict"],
                    help="f64 (default, the reported arithmetic), f32 (throughput mode), f64strict (RTTNW_F64_STRICT: nothing contracted)")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (per GPU)")
    ap.add_argument("--size", type=int, default=0, help="override width = height")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--counter-spp", type=int, default=8)
    ap.add_argument("--counter-level", type=int, default=1, help="collect_counters of the untimed counting pass (2, 3: more RTTNW_DEBUG_SCHED statistics)")
    ap.add_argument("--no-other", action="store_true", help="skip the timing of the other precision's kernels")
    ap.add_argument("--no-sub", action="store_true", help="skip the cornell_box / spheres_1m sub-records of the default run")
    ap.add_argument("--sub-steps", type=int, default=3, help="timed steps of each sub-record (after one warm-up step)")
    ap.add_argument("--spp-chunk", type=int, default=0, help="samples per work item (0 = the library's tapered schedule)")
    ap.add_argument("--bvh", default="auto", choices=["auto", "sah", "lbvh", "dsah"],
                    help="BVH builder at commit: auto (the library's default: host binned SAH below 100 000 leaves per tree, device binned SAH from there), host binned SAH, device LBVH, device binned SAH")
    ap.add_argument("--dump-image", default=None, metavar="PATH.npz", help="rank 0 saves the last timed step's frame (linear, rgba8): tests compare it with a single-rank render")
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
    # RTTNW_BENCH_ONE_DEVICE=1: rehearsal of an N-rank run on a box with ONE GPU — every rank on device 0, gloo instead of RCCL (which
    # refuses two ranks on one device) for the barriers, the max-over-ranks and the gather; everything else is the N-GPU code path
    one_device = os.environ.get("RTTNW_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
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
            if one_device:
                dist.init_process_group("gloo")
            else:
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
    earth = S.load_earth()

    PRECISION_NAMES = {abi.F32: "f32", abi.F64: "f64", abi.F64_STRICT: "f64strict"}

    class Workload:
        """One named workload resident on the device: scene built and committed, timing and counting helpers."""

        def __init__(self, workload, spp_override=0, size_override=0):
            self.workload = workload
            self.scene_name, self.W, self.H, spp1, self.param = WORKLOADS[workload]
            if size_override:
                self.W = self.H = size_override
            self.spp1 = spp_override or spp1
            self.spp = self.spp1 * world  # weak scaling: fixed per-GPU work
            t0 = time.time()
            self.sc, self.setup = S.build(gpu, scenes, self.scene_name, earth, self.param,
                                          bvh={"sah": abi.BVH_HOST_SAH, "lbvh": abi.BVH_DEVICE_LBVH, "dsah": abi.BVH_DEVICE_SAH}.get(args.bvh))
            self.build_s = time.time() - t0
            self.binfo = self.sc.build_info()
            self.info = abi.Stats()
            gpu.scene_info(self.sc.handle, self.info)
            self.samples_total = float(self.W) * self.H * self.spp          # all ranks together
            self.samples_rank = self.samples_total / world
            if share is not None:
                self.samples_total = self.samples_rank                       # --share: this rank's rate
            # (the committed PMC summaries are of the default configuration: whole frame on one GPU, host SAH trees, default schedule)
            self.default_run = (world == 1 and share is None and not spp_override and not size_override and not args.spp_chunk
                                and args.bvh == "auto")

        def counted(self, prec):
            """Counted rays / node visits / primitive tests / texels per sample (counted, not modelled), by the
            counting kernel variant on this rank's tiles (untimed)."""
            cam_c, pc = S.params_for(self.setup, self.W, self.H, args.counter_spp, precision=prec, tile_rank=rank, tile_world=world,
                                     seed=1, collect_counters=args.counter_level)
            rc_ = render.DeviceRenderer(self.sc, cam_c, pc)
            st = abi.Stats()
            rc_.trace(st)
            n = max(1, st.samples)
            per = dict(rays=st.rays / n, nodes=st.nodes_visited / n, prims=st.prims_tested / n, texels=st.texel_fetches / n)
            return per, st.reserved

        def timed(self, prec, steps, warmup):
            """`steps` timed steps after `warmup` untimed ones: (ms_per_step over the barrier-bracketed region, mean device
            time of the trace kernel(s) per step from the library's HIP events on the launch stream)."""
            cam_t, p = S.params_for(self.setup, self.W, self.H, self.spp, precision=prec, tile_rank=rank, tile_world=world, seed=1,
                                    spp_chunk=args.spp_chunk)
            r = render.DeviceRenderer(self.sc, cam_t, p)
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
            self.rank_kernel_ms = None
            if use_dist:
                t = torch.tensor([ms_per_step, kernel_ms], dtype=torch.float64, device="cpu" if one_device else "cuda")
                # every rank's own kernel time, for the spread (an imbalance of the tile partition would show here, not in the maximum)
                each = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
                dist.all_gather(each, t)
                ks = [float(e[1]) for e in each]
                self.rank_kernel_ms = {"min": round(min(ks), 3), "mean": round(sum(ks) / len(ks), 3), "max": round(max(ks), 3)}
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms_per_step, kernel_ms = float(t[0]), float(t[1])
            if args.dump_image and rank == 0 and share is None:
                torch.cuda.synchronize()
                np.savez(args.dump_image, linear=r.linear.cpu().numpy(), rgba8=r.rgba8.cpu().numpy())
                args.dump_image = None   # (the reported precision's frame only)
            del r
            return ms_per_step, kernel_ms

        def roofline(self, prec, kernel_ms):
            per, form = self.counted(prec)
            pname = PRECISION_NAMES[prec]
            secs = kernel_ms * 1e-3

            # SURVEY 8(d) to the letter: 32 B per node visit + 32 B per primitive test + 4 B per texel + 16 B / spp
            def alg_bytes(node_bytes):
                return node_bytes * per["nodes"] + 32.0 * per["prims"] + 4.0 * per["texels"] + 16.0 / self.spp

            def gbps(node_bytes):
                return alg_bytes(node_bytes) * self.samples_rank / secs / 1e9 if secs > 0 else 0.0

            b_alg = alg_bytes(NODE_VISIT_BYTES)
            hbm_alg = {"achieved": round(gbps(32.0), 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": round(gbps(32.0) / HBM_PEAK_GBPS, 5),
                       "alg_bytes_per_sample": round(b_alg, 2), "alg_bytes_per_launch": round(b_alg * self.samples_rank),
                       "frac_at_64B_per_node_visit": round(gbps(64.0) / HBM_PEAK_GBPS, 5),
                       "frac_at_128B_per_node_visit": round(gbps(128.0) / HBM_PEAK_GBPS, 5),
                       "note": "SURVEY 8(d): 32 B x node visits + 32 B x primitive tests + 4 B x texels + 16 B / spp, counts from the "
                               "counting kernel variant in this run; the 64 B reading prices a 4-wide record per child box (16 B), "
                               "the 128 B reading at what a visit physically reads"}
            pmc = committed_pmc(self.scene_name, pname) if self.default_run else None
            traffic = traffic_src = valu = None
            if pmc is not None and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                # KB -> B; FETCH_SIZE x 2: the gfx950 correction, calibrated on gathers of 128-B records (profiles/r03/README.md)
                traffic = round((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0)
                traffic_src = pmc["source"]
            if pmc is not None and pmc.get("SQ_INSTS_VALU") and pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU"):
                # The roof of a kernel bound by vector-instruction issue, from ITS OWN instruction mix: the launch's instructions by
                # class x the measured issue cycles of the class = the SIMD cycles the launch needs at best; peak = instructions per second
                # at that mean cost and the maximum clock with every lane of every instruction doing work.  achieved = instructions x lane
                # utilisation / time.  issue_utilisation = cycles needed / (1024 SIMDs x the launch's own GRBM_GUI_ACTIVE cycles): <= 1.
                n_valu = pmc["SQ_INSTS_VALU"]
                classified = {k: pmc[k] for k in VALU_COST_CYCLES if k in pmc}
                other = max(0.0, n_valu - sum(classified.values()))
                need = sum(v * VALU_COST_CYCLES[k] for k, v in classified.items()) + other * VALU_COST_OTHER_CYCLES
                mean_cost = need / n_valu
                lane_util = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0)
                peak = N_SIMDS * CLOCK_GHZ / mean_cost                                       # G wave-instructions / s the chip can issue of this mix
                ach = n_valu * lane_util / secs / 1e9 if secs > 0 else 0.0                   # full-wave equivalents / s
                issue_util = need / (N_SIMDS * pmc["GRBM_GUI_ACTIVE"] / 8.0) if pmc.get("GRBM_GUI_ACTIVE") else None
                valu = {"achieved": round(ach, 2), "peak": round(peak, 1), "unit": "G full-wave VALU instructions/s",
                        "frac": round(ach / peak, 5), "lane_utilisation": round(lane_util, 4),
                        "issue_utilisation": round(issue_util, 4) if issue_util is not None else None,
                        "wave_instructions_per_launch": n_valu,
                        "wave_instructions_per_sample": round(n_valu / self.samples_rank, 1),
                        "mean_issue_cycles_per_instruction": round(mean_cost, 3),
                        "instruction_classes": {k.replace("SQ_INSTS_VALU_", "").lower(): v for k, v in classified.items()},
                        "unclassified_instructions": other, "source": pmc["source"],
                        "note": "cycles the launch needs = sum over SQ_INSTS_VALU_* classes of count x measured issue cycles "
                                "(profiles/r04/valu_cost_cycles.txt); issue_utilisation = that / (1024 SIMDs x the launch's GRBM_GUI_ACTIVE cycles); "
                                "achieved = SQ_INSTS_VALU x SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) / kernel time: the launch's vector "
                                "instructions counted as full 64-lane instructions; peak = 1024 SIMDs x 2.4 GHz / the mix's mean issue cycles; "
                                "frac ~ issue_utilisation x lane_utilisation"}
            lds_resident = (form & 2) != 0   # rttnw_stats.reserved bit 1: the launch kept the node records in LDS (the library's own choice)
            node_steps = 3 if (form & 4) else 2   # bit 2: the instantiation with three node steps per walk trip (tiny top trees)
            # the instantiation that ran, as rocprofv3's kernel trace names it (<R, COUNT, BLOCK, LDS nodes, SHAPES, node steps per trip>; the bench
            # scenes have none of the rare graph shapes of the GENERAL instantiations)
            rname = "float" if prec == abi.F32 else "double"
            ns = "rt::ieee_strict" if prec == abi.F64_STRICT else "rt::contracted"   # (rt_core.hpp: the two builds of the arithmetic)
            shapes = (3 if (form & 16) else 2) if (form & 8) else 0   # bits 3, 4: the instantiations for walks that never change frames (rt_core.hpp SHAPES_NONE / SHAPES_SINGLE)
            if (form & 32) and shapes:
                shapes += 2                                               # bit 5: their LEAN flavours (SHAPES_NONE_NT = 4 / SHAPES_SINGLE_NT = 5: no moving sphere, no medium, solid colours)
            kernel = ("%s::trace_kernel<%s, false, %d>" % (ns, rname, shapes) if (form & 1) != 0 else
                      "%s::trace_kernel_plain<%s, false, %s, %d, %d>" % (ns, rname, "1024, true" if lds_resident else "256, false", shapes, node_steps))
            common = {"traffic": traffic, "traffic_source": traffic_src, "kernel": kernel, "kernel_ms": round(kernel_ms, 3),
                      "per_sample": {k: round(v, 3) for k, v in per.items()},
                      "grays_per_s": round(per["rays"] * self.samples_rank / secs / 1e9, 3) if secs > 0 else None,  # world.hit() calls / s
                      "hbm_algorithmic": hbm_alg, "valu": valu}
            if traffic is not None and secs > 0:
                common["traffic_frac_of_hbm_peak"] = round(traffic / secs / 1e9 / HBM_PEAK_GBPS, 5)
            # the scalars a reader of the line wants first, flat (nested objects may be dropped by whoever parses the line): SURVEY 8(d)'s byte
            # fraction whatever `bound` says, and — while a PMC summary of this kernel source is committed — what the counters say beside it
            common["hbm_alg_frac"] = hbm_alg["frac"]
            common["hbm_alg_gbps"] = hbm_alg["achieved"]
            common["lane_utilisation"] = valu["lane_utilisation"] if valu else None
            common["issue_utilisation"] = valu["issue_utilisation"] if valu else None
            common["valu_insts_per_sample"] = valu["wave_instructions_per_sample"] if valu else None
            common["traffic_over_algorithmic"] = round(traffic / hbm_alg["alg_bytes_per_launch"], 4) if traffic is not None and hbm_alg["alg_bytes_per_launch"] else None
            common["alg_bytes_per_sample"] = hbm_alg["alg_bytes_per_sample"]
            # where the counter-derived fields (traffic, lane / issue utilisation) come from: PMC counters exist only under rocprofv3, so they are
            # the COMMITTED summary of this kernel source (hash-gated, collected on the builder's lease) combined with THIS run's kernel_ms
            common["pmc"] = "committed" if pmc is not None else None
            # what bounds the kernel: memory when the node records are walked in HBM / Infinity Cache (the decoupled kernel:
            # counter traffic ~ algorithmic bytes), vector-instruction issue when they are LDS-resident (counter traffic ~1 % of peak)
            if lds_resident and valu is not None:
                out = {"bound": "valu", "achieved": valu["achieved"], "peak": valu["peak"], "unit": valu["unit"], "frac": valu["frac"]}
                out["note"] = ("node records are LDS-resident: the kernel is bound by vector-instruction issue at this lane utilisation, "
                               "not by memory (see traffic); hbm_algorithmic carries SURVEY 8(d)'s byte accounting for the same launch")
            else:
                out = {"bound": "hbm", "achieved": hbm_alg["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_alg["frac"]}
                out["note"] = hbm_alg["note"] + ("" if not lds_resident else
                                                 "; node records are LDS-resident and the kernel is VALU-issue bound — no PMC summary of "
                                                 "this kernel source is committed, so only the byte accounting is reported")
            out.update(common)
            return out

        def record(self, prec, steps, warmup):
            ms, kms = self.timed(prec, steps, warmup)
            return {"workload": "%s %dx%d spp=%d" % (self.scene_name, self.W, self.H, self.spp),
                    "dtype": "f32" if prec == abi.F32 else "f64", "precision": PRECISION_NAMES[prec],
                    "value": round(self.samples_total / (ms * 1e-3) / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(ms, 3),
                    "steps": steps, "warmup": warmup, "roofline": self.roofline(prec, kms)}

        def config(self):
            return {"scene_nodes": self.info.n_nodes, "scene_prims": self.info.n_prims, "scene_bytes_f32": self.info.scene_bytes,
                    "scene_build_s": round(self.build_s, 3), "bvh_builder": args.bvh, "bvh_lower_ms": round(self.binfo.lower_ms, 2),
                    "bvh_device_ms": round(self.binfo.device_ms, 3), "stack_depth": self.binfo.stack_depth}

    workload = args.workload or ("final_scene" if world == 1 else "final_scene_1600")
    wl = Workload(workload, args.spp, args.size)
    precision = {"f32": abi.F32, "f64": abi.F64, "f64strict": abi.F64_STRICT}[args.precision]

    # ---- N > 1: what ONE GPU does with the same frame at the per-GPU sample count — rank 0 alone traces the whole 1600x1600 frame at
    # spp = 1250 (one step after one warm-up; the other ranks wait at the barrier), so that the line carries its own single-GPU reference on
    # the same frame (the N = 1 bench line is the 800x800 headline frame: ~1.7 % apart per GPU) and the efficiency read off it is not inflated
    per_gpu_reference = None
    if use_dist and world > 1:
        assert dist.get_world_size() == world == args.gpus, "bench: %d ranks joined, WORLD_SIZE %d, --gpus %d" % (dist.get_world_size(), world, args.gpus)
        if rank == 0:
            cam1, p1 = S.params_for(wl.setup, wl.W, wl.H, wl.spp1, precision=precision, tile_rank=0, tile_world=1, seed=1, spp_chunk=args.spp_chunk)
            r1 = render.DeviceRenderer(wl.sc, cam1, p1)
            r1.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            st1 = abi.Stats()
            r1.trace(st1)
            r1.collect()
            torch.cuda.synchronize()
            ms1 = (time.perf_counter() - t1) * 1e3
            per_gpu_reference = {"workload": "%s %dx%d spp=%d on ONE GPU (rank 0 alone, 1 step)" % (wl.scene_name, wl.W, wl.H, wl.spp1),
                                 "value": round(float(wl.W) * wl.H * wl.spp1 / (ms1 * 1e-3) / 1e6, 3), "unit": "Msamples/s",
                                 "ms_per_step": round(ms1, 3), "kernel_ms": round(st1.kernel_ms, 3)}
            del r1
        dist.barrier()

    # ---- timed region (the reported precision)
    ms_per_step, kernel_ms = wl.timed(precision, args.steps, args.warmup)
    rank_kernel_ms = wl.rank_kernel_ms
    value = wl.samples_total / (ms_per_step * 1e-3) / 1e6
    roof = wl.roofline(precision, kernel_ms)

    # ---- the same workload through the other precision's kernels, its own multi-step timing; and through the IEEE-strict build of the f64
    # kernels (RTTNW_F64_STRICT: nothing contracted, every object tested in the reference's frame — the build whose pixels ARE the CPU
    # reference's, tests/test_gpu_parity.py::test_f64_strict_takes_the_oracles_decisions)
    other = strict = None
    if not args.no_other:
        oprec = abi.F64 if precision == abi.F32 else abi.F32
        other = wl.record(oprec, max(1, min(args.steps, 5)), 1)
        if precision != abi.F64_STRICT and world == 1 and share is None:
            strict = wl.record(abi.F64_STRICT, max(1, min(args.steps, 5)), 1)

    def cpu_sample(w, seconds, name):
        """The CPU oracle on the host cores for a bounded sample of workload `w` at its full frame and geometry: calibrate at 1 spp, then
        the spp that fills `seconds`.  spheres_1m's 10^6 spheres go through the oracle's median-split builder (the reference's own
        builder is O(n^2 log n): hittable.rs:265-321; tests/test_oracle_kat.py proves the two bit-identical where both can run)."""
        from oracle import rto
        big = w.scene_name == "spheres_1m"
        so, _ = S.build(rto.binding(), scenes, w.scene_name, earth, w.param, bvh=rto.BVH_MEDIAN_SPLIT if big else None)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cal = 1
        while True:                                                        # calibration at full size: enough spp for >= 0.4 s of work
            camc, pcal = S.params_for(w.setup, w.W, w.H, cal, seed=1)
            tc = time.perf_counter()
            rto.render(so, camc, pcal, n_threads=cores, want_rgba8=False)
            dt = time.perf_counter() - tc
            if dt >= 0.4 or cal >= 64:
                break
            cal = min(64, max(2 * cal, int(cal * 0.6 / max(dt, 1e-3))))
        rate = w.W * w.H * cal / dt
        cspp = int(max(1, min(256, round(rate * seconds / (w.W * w.H)))))
        camc, pcpu = S.params_for(w.setup, w.W, w.H, cspp, seed=1)
        tc = time.perf_counter()
        rto.render(so, camc, pcpu, n_threads=cores, want_rgba8=False)
        dt = time.perf_counter() - tc
        return {"value": round(w.W * w.H * cspp / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                "sample": "%s %dx%d spp=%d (%.1f s), f64 CPU oracle (reference-shaped: list scan + %s)"
                          % (w.scene_name, w.W, w.H, cspp, dt, "BvhTree::hit over a median-split tree: the reference's builder cannot make this one" if big
                             else "the reference's BVH builder")}

    # ---- the other single-GPU configs of BASELINE.json, timed in the same run (default N = 1 run only).  spheres_1m reports the
    # IEEE-strict f64 build first: on that scene (rounding grows ~100x per bounce) it is the build whose pixels equal the CPU
    # reference's (tests/test_gpu_parity.py::test_config5_spheres_1m_at_its_size_vs_oracle); the contracted build rides along.
    subs, subs_detail = {}, {}
    if world == 1 and share is None and args.workload is None and not args.no_sub and not args.spp and not args.size:
        for name in ("cornell_box", "spheres_1m"):
            w2 = Workload(name)
            recs = {pn: w2.record(pr, args.sub_steps, 1) for pn, pr in (("f64strict", abi.F64_STRICT), ("f64", abi.F64), ("f32", abi.F32))}
            cpu2 = cpu_sample(w2, min(args.cpu_seconds, 6.0), name) if rank == 0 and args.cpu_seconds > 0 else None
            subs[name] = {"workload": recs["f64"]["workload"]}
            subs[name].update({pn: flat_sub(rec, cpu2 if pn == "f64" else None) for pn, rec in recs.items()})
            subs_detail[name] = dict(recs, config=w2.config(), cpu_baseline=cpu2)
            del w2

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu = cpu_sample(wl, args.cpu_seconds, workload)

    if rank == 0 or share is not None:
        cfg = {"workload": "%s %dx%d spp=%d%s" % (wl.scene_name, wl.W, wl.H, wl.spp, " (spp = %d x %d GPUs)" % (wl.spp1, world) if world > 1 else ""),
               "max_depth": 50, "scene_seed": "0x5eed0001", "render_seed": 1, "quirks": "reference",
               "partition": ("8x8 tiles interleaved over %d rank(s), RCCL gather to rank 0" % world) if share is None else
                            ("--share: rank %d of %d only, on one GPU, no gather; value = this rank's rate" % (rank, world))}
        cfg.update(wl.config())
        out = {
            "metric": "Msamples/sec on final_scene 800x800 spp=1000; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": 1 if share is not None else world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == abi.F32 else "f64", "precision": args.precision, "data": "synthetic",
            "config": cfg,
            "roofline": flat_roofline(roof),
            "cpu_baseline": cpu,
        }
        detail = dict(out, roofline=roof)
        # the headline workload through the other builds, flat: the IEEE-strict build first — its pixels ARE the CPU reference's (tier bar 1)
        if strict is not None:
            out.update(strict_value=strict["value"], strict_ms_per_step=strict["ms_per_step"], strict_kernel_ms=strict["roofline"]["kernel_ms"],
                       strict_steps=strict["steps"], strict_hbm_alg_frac=strict["roofline"]["hbm_alg_frac"],
                       strict_lane_utilisation=strict["roofline"]["lane_utilisation"])
            detail["f64strict_kernels"] = strict
        if other is not None:
            on = "f64" if precision == abi.F32 else "f32"
            out.update({on + "_value": other["value"], on + "_ms_per_step": other["ms_per_step"], on + "_kernel_ms": other["roofline"]["kernel_ms"],
                        on + "_steps": other["steps"]})
            detail[on + "_kernels"] = other
        if use_dist:
            out["distributed"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks_expected": args.gpus,
                                  "rank_kernel_ms": rank_kernel_ms}
            detail["distributed"] = out["distributed"]
        if per_gpu_reference is not None:
            # (the driver computes its own efficiency from the N = 1, 2, 4, 8 lines; this one is against the SAME frame on one GPU of this run)
            per_gpu_reference["scaling_efficiency"] = round(value / (world * per_gpu_reference["value"]), 4)
            out["per_gpu_reference"] = detail["per_gpu_reference"] = per_gpu_reference
        if subs:
            out["sub"] = subs
            detail["sub"] = subs_detail
        line = json.dumps(out, separators=(",", ":"))
        # (a line beyond the limit must never cost the run its measurement: the optional parts go first — they are in the detail file anyway)
        for optional in ("sub", "distributed", "per_gpu_reference"):
            if len(line) <= LINE_LIMIT:
                break
            log("bench: the stdout line is %d bytes (limit %d): dropping '%s' from it (see the detail file)" % (len(line), LINE_LIMIT, optional))
            out.pop(optional, None)
            line = json.dumps(out, separators=(",", ":"))
        dpath = os.environ.get("RTTNW_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(dpath), exist_ok=True)
            with open(dpath, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            log("bench: detail file not written (%s)" % e)
        log("bench detail: " + json.dumps(detail))
        print(line, flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
