#!/bin/bash
# rocprofv3 evidence for the progressive / changed_box kernels (fill_pass_kernel, fill_pass_quad_kernel, fill_pass_rows_kernel):
# kernel trace + stats, then separate PMC passes (WRITE_SIZE, FETCH_SIZE, SQ) -- no tracing domain besides --kernel-trace.
# usage: tools/gpu_profile_pass.sh <tag> [side]
TAG=${1:-pass}
SIDE=${2:-256}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python tools/pass_workload.py $SIDE"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wr -o pmc --output-format csv -- $CMD > $OUT/pmc_wr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o pmc --output-format csv -- $CMD > $OUT/pmc_rd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc --output-format csv -- $CMD > $OUT/pmc_sq.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $CMD   (tools/pass_workload.py: 10 repetitions of every progressive case at ${SIDE}^3)"
  cat $OUT/trace/trace_kernel_stats.csv
  echo
  echo "# per (kernel, grid size)"
  python tools/kernel_trace_avg.py $OUT/trace
  echo
  echo "# PMC passes (one rocprofv3 --pmc run each), per (kernel, grid size): WRITE_SIZE / FETCH_SIZE in KiB (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md)"
  python tools/pmc_by_grid.py $OUT
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
