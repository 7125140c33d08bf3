#!/bin/bash
# Round-2 profile collection (run on the GPU box from the repo root: bash profiles/collect_r02.sh): bench lines, rocprofv3
# --kernel-trace --stats per kernel, PMC summaries (separate --pmc passes, profiles/collect_pmc.sh).  Output under
# gpurun_out/r02/; the summaries judged are copied into profiles/r02/ (see profiles/r02/README.md).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
python bench.py --steps 10 --warmup 2 > $O/bench_r02.json 2> $O/bench_r02.err; cat $O/bench_r02.json
python bench.py --workload cornell_box --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_cornell_box.json 2>/dev/null
python bench.py --workload spheres_1m --steps 5 --warmup 1 --cpu-seconds 0 > $O/bench_spheres_1m.json 2>/dev/null
python bench.py --workload spheres_1m --bvh lbvh --steps 3 --warmup 1 --cpu-seconds 0 --no-other --precision f32 > $O/bench_spheres_1m_lbvh_f32.json 2>/dev/null
cat $O/bench_cornell_box.json $O/bench_spheres_1m.json $O/bench_spheres_1m_lbvh_f32.json | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); o=d.get('f32_kernels') or d.get('f64_kernels') or {}
    print(d['config']['workload'], d['dtype'], d['value'], d['roofline']['frac'], '| other', o.get('value'), (o.get('roofline') or {}).get('frac'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_f64 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-other > $GRAFT_REPO_ROOT/$O/kt_f64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_f32 -- python3 $GRAFT_REPO_ROOT/bench.py --precision f32 --steps 4 --warmup 1 --cpu-seconds 0 --no-other > $GRAFT_REPO_ROOT/$O/kt_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_s1m -- python3 $GRAFT_REPO_ROOT/bench.py --workload spheres_1m --precision f32 --steps 3 --warmup 1 --cpu-seconds 0 --no-other > $GRAFT_REPO_ROOT/$O/kt_s1m.log 2>&1
cd $GRAFT_REPO_ROOT
bash profiles/collect_pmc.sh $O/pmc_final_scene_f64 > $O/pmc_f64.log 2>&1
bash profiles/collect_pmc.sh $O/pmc_final_scene_f32 --precision f32 > $O/pmc_f32.log 2>&1
bash profiles/collect_pmc.sh $O/pmc_spheres_1m_f32 --workload spheres_1m --precision f32 > $O/pmc_s1m.log 2>&1
for d in pmc_final_scene_f64 pmc_final_scene_f32 pmc_spheres_1m_f32; do echo "== $d"; grep -E "lane_util|valu_busy|wave_time|FETCH_SIZE|WRITE_SIZE|VGPR|kernel_source" $O/$d/summary.txt; done
