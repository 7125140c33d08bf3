#!/bin/bash
# Round-4 profile collection (one gpurun call, repo root): the default bench line, rocprofv3 kernel statistics and PMC summaries of the
# dominant kernel of each workload x precision, phase breakdowns.  Outputs under gpurun_out/r04 (copy what is to be judged to profiles/r04).
OUT=gpurun_out/r04
mkdir -p $OUT
ROOT=$(pwd)
for spec in "final_scene f64" "final_scene f32" "cornell_box f64" "spheres_1m f64strict" "spheres_1m f64" "spheres_1m f32"; do
  set -- $spec
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats_$1_$2 -- python3 $ROOT/bench.py --workload $1 --precision $2 --steps 4 --warmup 1 --cpu-seconds 0 --no-other --no-sub > $ROOT/$OUT/stats_$1_$2.log 2>&1 )
  f=$(find $OUT/stats_$1_$2 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$1_$2.csv
done
for spec in "final_scene f64" "final_scene f32" "cornell_box f64" "cornell_box f32" "spheres_1m f64strict" "spheres_1m f64" "spheres_1m f32"; do
  set -- $spec
  bash profiles/collect_pmc.sh $OUT/pmc_$1_$2 --workload $1 --precision $2 > /dev/null 2>&1
  cp $OUT/pmc_$1_$2/summary.txt $OUT/pmc_$1_$2.txt
done
bash profiles/collect_phases.sh $OUT
# the default bench line LAST, with this run's PMC summaries in place (bench.py reads roofline.valu / traffic from profiles/r04 while their
# kernel_source_sha is the tree's)
cp $OUT/pmc_*.txt profiles/r04/
python3 bench.py --steps 10 --warmup 2 > $OUT/bench_r04.json 2> $OUT/bench_r04.err
ls $OUT
