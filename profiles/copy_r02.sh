#!/bin/bash
# After `gpurun -- bash profiles/collect_r02.sh`: copy the summaries that are judged from gpurun_out/r02/ (scratch, merged back
# by gpurun) into profiles/r02/ (tracked).  Run from the repo root; then update the numbers quoted in profiles/r02/README.md.
set -e
O=gpurun_out/r02; P=profiles/r02
cp $O/bench_r02.json $O/bench_cornell_box.json $O/bench_spheres_1m.json $O/bench_spheres_1m_lbvh_f32.json $P/
for k in f64 f32 s1m; do
  f=$(ls -t $O/kt_$k/*/*_kernel_stats.csv | head -1)
  case $k in s1m) n=spheres_1m_f32;; *) n=final_scene_$k;; esac
  cp "$f" $P/kernel_stats_$n.csv
done
for d in final_scene_f64 final_scene_f32 spheres_1m_f32; do cp $O/pmc_$d/summary.txt $P/pmc_$d.txt; done
python3 - <<'PY'
import csv, json
P = "profiles/r02/"
for f in ("bench_r02", "bench_cornell_box", "bench_spheres_1m", "bench_spheres_1m_lbvh_f32"):
    d = json.load(open(P + f + ".json")); o = d.get("f32_kernels") or {}
    print(f, d["dtype"], d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"],
          "| f32", o.get("value"), (o.get("roofline") or {}).get("frac"))
for f in ("final_scene_f64", "final_scene_f32", "spheres_1m_f32"):
    for r in csv.DictReader(open(P + "kernel_stats_%s.csv" % f)):
        if "trace_kernel" in r["Name"] and ", false" in r["Name"][:60]:
            print(f, r["Calls"], "calls, average %.1f ms" % (float(r["AverageNs"]) / 1e6))
PY
