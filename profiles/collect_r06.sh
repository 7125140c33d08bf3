#!/bin/bash
# Round-6 profile collection (one gpurun call, repo root): rocprofv3 kernel statistics and PMC summaries of the dominant kernel of each
# workload x precision (RTTNW_F64_STRICT included for all three workloads), phase breakdowns of the counting variants (which mirror the
# asynchronous kernel since this round), then the default bench line with this run's PMC summaries in place.  Outputs under gpurun_out/r06
# (copy what is to be judged to profiles/r06).
OUT=gpurun_out/r06
mkdir -p $OUT profiles/r06
ROOT=$(pwd)
for spec in "final_scene f64" "final_scene f64strict" "final_scene f32" "cornell_box f64" "cornell_box f64strict" "spheres_1m f64strict" "spheres_1m f64" "spheres_1m f32"; do
  set -- $spec
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats_$1_$2 -- python3 $ROOT/bench.py --workload $1 --precision $2 --steps 4 --warmup 1 --cpu-seconds 0 --no-other --no-sub > $ROOT/$OUT/stats_$1_$2.log 2>&1 )
  f=$(find $OUT/stats_$1_$2 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$1_$2.csv
done
for spec in "final_scene f64" "final_scene f64strict" "final_scene f32" "cornell_box f64" "cornell_box f64strict" "cornell_box f32" "spheres_1m f64strict" "spheres_1m f64" "spheres_1m f32"; do
  set -- $spec
  bash profiles/collect_pmc.sh $OUT/pmc_$1_$2 --workload $1 --precision $2 > /dev/null 2>&1
  cp $OUT/pmc_$1_$2/summary.txt $OUT/pmc_$1_$2.txt
done
bash profiles/collect_phases.sh $OUT
cp $OUT/pmc_*.txt profiles/r06/
RTTNW_BENCH_DETAIL=$OUT/bench_detail_r06.json python3 bench.py --steps 20 --warmup 5 > $OUT/bench_r06.json 2> $OUT/bench_r06.err
ls $OUT
