#!/bin/bash
# usage: gpu_pmc.sh <tag> "<counters...>" [workload]   -- one PMC pass of bench.py (no tracing domains)
TAG=$1; CTRS=$2; WL=${3:-256}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d $OUT/pmc_x -o pmc --output-format csv -- python bench.py --steps 10 --warmup 2 --workload $WL --no-cpu-baseline > $OUT/pmc_x.log 2>&1
python tools/summarize_pmc.py $OUT
