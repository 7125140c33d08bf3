#!/bin/bash
# Phase breakdown of the trace kernels (the counting variants' wave clocks and lane tallies, RTTNW_DEBUG_SCHED=1; debug_sched.cpp prints
# them): one file per workload x precision under <outdir>.  Usage (GPU box, repo root): bash profiles/collect_phases.sh profiles/r04
OUT=${1:-gpurun_out/phases}
mkdir -p $OUT
for w in final_scene cornell_box spheres_1m; do
  for p in f64 f64strict f32; do
    for lvl in 1 2 3; do
      RTTNW_DEBUG_SCHED=1 python3 bench.py --workload $w --precision $p --steps 1 --warmup 0 --cpu-seconds 0 --no-other --counter-spp 256 --counter-level $lvl 2> $OUT/phases_${w}_${p}_l$lvl.txt > /dev/null
    done
    cat $OUT/phases_${w}_${p}_l1.txt $OUT/phases_${w}_${p}_l2.txt $OUT/phases_${w}_${p}_l3.txt | grep -E "^\[(plain|decoupled)\]" > $OUT/phases_${w}_${p}.txt
    rm -f $OUT/phases_${w}_${p}_l1.txt $OUT/phases_${w}_${p}_l2.txt $OUT/phases_${w}_${p}_l3.txt
  done
done
