#!/bin/bash
# rocprofv3 evidence for the progressive / changed_box kernels: per GROUP of cases (tools/pass_workload.py: one group per
# process, warm, 100 repetitions) a kernel trace + stats and two separate PMC passes (WRITE_SIZE, FETCH_SIZE) -- no tracing
# domain besides --kernel-trace -- reduced to profiles/<tag>_pass_traffic.json by tools/pass_traffic.py.
# usage: tools/gpu_profile_pass.sh <tag> [side]
TAG=${1:-pass}
SIDE=${2:-256}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
for G in load_virgin load_unflagged fresh_step1 edit_full edit_eighth noop; do
  CMD="python tools/pass_workload.py $SIDE $G"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$G/trace -o trace --output-format csv -- $CMD > $OUT/$G.trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/$G/pmc_wr -o pmc --output-format csv -- $CMD > $OUT/$G.pmc_wr.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$G/pmc_rd -o pmc --output-format csv -- $CMD > $OUT/$G.pmc_rd.log 2>&1
done
python tools/pass_traffic.py $OUT $SIDE > $OUT/pass_traffic_$SIDE.json 2> $OUT/pass_traffic.err
{
  python tools/source_hash.py --stamp
  echo "# rocprofv3 --kernel-trace --stats / --pmc WRITE_SIZE / --pmc FETCH_SIZE -- python tools/pass_workload.py $SIDE <group>"
  echo "# one group of cases per process, 0.3 s of pre-warm, 100 repetitions (PASS_REPS); per (kernel, grid size)"
  for G in load_virgin load_unflagged fresh_step1 edit_full edit_eighth noop; do
    echo; echo "## group $G"
    python tools/kernel_trace_avg.py $OUT/$G/trace 20
    python tools/pmc_by_grid.py $OUT/$G | grep -v "vectorized_elementwise\|^==" 
  done
} > $OUT/summary.txt 2>&1
cat $OUT/pass_traffic_$SIDE.json
