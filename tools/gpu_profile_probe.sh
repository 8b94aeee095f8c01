#!/bin/bash
# The run the bench LINE describes, under the profiler: default bench.py options (all three pipelines), kernel trace + stats;
# tools/kernel_trace_avg.py lists the launches per (kernel, grid size).  usage: tools/gpu_profile_probe.sh <tag> [workload]
TAG=${1:-probe}
WL=${2:-256}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 3 --workload $WL --no-cpu-baseline --no-batch --no-overlapped"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $CMD   (the driver's default command: the line bench.py prints describes this run)"
  python tools/bench_line_of.py $OUT/trace.log
  cat $OUT/trace/trace_kernel_stats.csv
  echo
  echo "# per (kernel, grid size)"
  python tools/kernel_trace_avg.py $OUT/trace
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
