#!/bin/bash
# Round-3 profile collection (run on the GPU box from the repo root: bash profiles/collect_r03.sh): the default bench line,
# rocprofv3 --kernel-trace --stats per workload, PMC summaries (separate --pmc passes, profiles/collect_pmc.sh).  Output under
# gpurun_out/r03/; the summaries judged are copied into profiles/r03/ (profiles/copy_r03.sh; see profiles/r03/README.md).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
# PMC first: the bench line reads `traffic` / `valu` from profiles/r03/pmc_*.txt while their kernel_source_sha matches
for cfg in "final_scene f64" "final_scene f32" "cornell_box f64" "cornell_box f32" "spheres_1m f32" "spheres_1m f64"; do
  set -- $cfg
  PMC_QUICK=${PMC_QUICK_ALL:-} bash profiles/collect_pmc.sh $O/pmc_$1_$2 --workload $1 --precision $2 > $O/pmc_$1_$2.log 2>&1
  cp $O/pmc_$1_$2/summary.txt profiles/r03/pmc_$1_$2.txt
  echo "== $cfg"; grep -E "lane_util|valu_busy|wave_time|FETCH_SIZE|WRITE_SIZE|VGPR|kernel_source|SQ_INSTS_VALU " $O/pmc_$1_$2/summary.txt
done
cp profiles/r03/pmc_*.txt $O/
python bench.py --steps 10 --warmup 2 > $O/bench_r03.json 2> $O/bench_r03.err; cat $O/bench_r03.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['roofline']['bound'], d['roofline']['frac'], 'hbm_alg', d['roofline']['hbm_algorithmic']['frac'], 'f32', d['f32_kernels']['value'], 'cpu', d['cpu_baseline']['value'])
for k in ('cornell_box','spheres_1m'):
    r=d[k]; print(k, r['value'], r['roofline']['bound'], r['roofline']['frac'], '| f32', r['f32_kernels']['value'], r['f32_kernels']['roofline']['bound'], r['f32_kernels']['roofline']['frac'], r['f32_kernels']['roofline'].get('traffic_frac_of_hbm_peak'))"
python bench.py --workload spheres_1m --bvh lbvh --steps 3 --warmup 1 --cpu-seconds 0 --no-other --no-sub --precision f32 > $O/bench_spheres_1m_lbvh_f32.json 2>/dev/null
python bench.py --workload spheres_1m --bvh dsah --steps 3 --warmup 1 --cpu-seconds 0 --no-other --no-sub --precision f32 > $O/bench_spheres_1m_dsah_f32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_f64 -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-other --no-sub > $R/$O/kt_f64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_f32 -- python3 $R/bench.py --precision f32 --steps 4 --warmup 1 --cpu-seconds 0 --no-other --no-sub > $R/$O/kt_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_cornell -- python3 $R/bench.py --workload cornell_box --steps 4 --warmup 1 --cpu-seconds 0 --no-other > $R/$O/kt_cornell.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt_s1m -- python3 $R/bench.py --workload spheres_1m --precision f32 --steps 3 --warmup 1 --cpu-seconds 0 --no-other > $R/$O/kt_s1m.log 2>&1
cd $R
for d in kt_f64 kt_f32 kt_cornell kt_s1m; do f=$(ls $O/$d/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== $d"; head -4 $f | cut -c1-200; cp $f $O/kernel_stats_$d.csv; done
python profiles/scenes_table.py > $O/scenes_table.md 2>/dev/null; cat $O/scenes_table.md
