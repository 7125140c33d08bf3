#!/bin/bash
# Collect PMC counters for the trace kernel in separate rocprofv3 passes (never combined with trace domains
# other than --kernel-trace).  Usage (on the GPU box, from the repo root):  bash profiles/collect_pmc.sh <outdir> [bench args]
OUT=${1:-gpurun_out/pmc}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-other --no-sub $BENCH_ARGS > $ROOT/$OUT/$name.log 2>&1
}
BENCH_ARGS="$*"
if [ -n "$PMC_QUICK" ]; then
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
else
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_BRANCH
run sq4 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC
run sq5 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
fi
# summarise: counter values of the trace kernel dispatches
python3 - "$ROOT/$OUT" "$ROOT" <<'PY'
import csv, glob, os, re, sys, collections
out = sys.argv[1]
sys.path.insert(0, sys.argv[2])
import bench
summ = collections.OrderedDict()
summ["kernel_source_sha"] = [bench.kernel_source_sha()]   # bench.py reports `traffic` from this file only while the kernel sources are these
for d in sorted(glob.glob(os.path.join(out, "*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" not in r.get("Kernel_Name", ""): continue
            if re.search(r"<(float|double), true", r["Kernel_Name"]): continue   # skip the counting variant (untimed pre-pass)
            key = r["Counter_Name"]
            summ.setdefault(key, []).append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_kernel" in r["Kernel_Name"] and not re.search(r"<(float|double), true", r["Kernel_Name"]):
                summ.setdefault("duration_ns[%s]" % os.path.basename(d), []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                summ["VGPR/SGPR/LDS/scratch/grid/wg"] = ["%s/%s/%s/%s/%s/%s" % (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size_X"], r["Workgroup_Size_X"])]
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k, v in summ.items():
        line = "%-40s %s" % (k, v[0] if isinstance(v[0], str) else "%.6g (n=%d)" % (sum(v) / len(v), len(v)))
        print(line); fo.write(line + "\n")
    g = lambda k: sum(summ[k]) / len(summ[k]) if k in summ else None
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU") and g("GRBM_GUI_ACTIVE"):
        # SIMD cycles the launch needs at the measured per-class issue costs (bench.py VALU_COST_CYCLES, profiles/r04/valu_cost_cycles.txt)
        # against the cycles it had: cannot exceed 1
        n_valu = g("SQ_INSTS_VALU")
        classified = {k: g(k) for k in bench.VALU_COST_CYCLES if g(k) is not None}
        need = sum(v * bench.VALU_COST_CYCLES[k] for k, v in classified.items()) + max(0.0, n_valu - sum(classified.values())) * bench.VALU_COST_OTHER_CYCLES
        d = ["lane_utilisation = SQ_THREAD_CYCLES_VALU/(SQ_ACTIVE_INST_VALU*64) = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)),
             "issue_utilisation = sum(class count x measured issue cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE/8) = %.3f  (mean %.2f cycles per instruction)" % (need / (bench.N_SIMDS * g("GRBM_GUI_ACTIVE") / 8), need / n_valu),
             "wave_time: wait_any %.3f  wait_inst %.3f  active %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"))]
        for line in d:
            print(line); fo.write(line + "\n")
PY
