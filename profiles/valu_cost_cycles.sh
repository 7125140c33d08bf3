#!/bin/bash
# Cycles per wave64 vector instruction and SIMD, by kind, from GRBM_GUI_ACTIVE of one long launch per kind (profiles/valu_cost.hip "pmc" mode):
# the unit the kernels' PMC summaries are in.  Usage (GPU box, repo root): bash profiles/valu_cost_cycles.sh > profiles/r04/valu_cost_cycles.txt
ROOT=$(pwd)
hipcc -O3 --offload-arch=gfx950 profiles/valu_cost.hip -o /tmp/valu_cost || exit 1
OUT=/tmp/valu_cost_pmc
rm -rf $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT -- /tmp/valu_cost pmc > /dev/null 2>&1 )
python3 - $OUT <<'PY'
import csv, glob, os, re, sys
names = {0: "v_fma_f32", 15: "v_max3_f32", 11: "v_cmp_lt_f32", 14: "v_cvt_f32_ubyte1", 13: "v_rcp_f32", 7: "v_xor_b32", 8: "v_cndmask_b32", 18: "v_add_co_u32",
         19: "v_bcnt_u32_b32", 4: "v_mul_lo_u32", 5: "v_mul_hi_u32", 6: "v_mad_u64_u32", 9: "v_lshrrev_b64", 1: "v_fma_f64", 2: "v_add_f64", 3: "v_mul_f64",
         10: "v_cmp_lt_f64", 16: "v_cvt_f64_u32", 12: "v_rcp_f64", 17: "v_sqrt_f64", 20: "v_fma_mix_f32", 21: "v_cvt_f32_f16", 22: "v_min_f32", 23: "v_mov_b32"}
cyc, dur = {}, {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"cost_kernel<(\d+)>", r["Kernel_Name"])
        if m and r["Counter_Name"] == "GRBM_GUI_ACTIVE": cyc[int(m.group(1))] = float(r["Counter_Value"]) / 8.0   # summed over the 8 XCDs
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"cost_kernel<(\d+)>", r["Kernel_Name"])
        if m: dur[int(m.group(1))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
per_simd = 8 * 32768 * 32.0   # instructions one SIMD issued: 8 waves x iterations x 32 per iteration
print("# cycles per wave64 instruction and SIMD at 8 waves per SIMD (GRBM_GUI_ACTIVE / 8 XCDs / instructions a SIMD issued), the launch's clock, ns")
for k, n in names.items():
    if k in cyc: print("%-18s %6.2f cycles   %.2f GHz   %.2f ns" % (n, cyc[k] / per_simd, cyc[k] / dur.get(k, float("nan")), dur.get(k, float("nan")) / per_simd))
PY
