#!/bin/bash
# Profiles `bench.py` on the GPU box: (1) kernel trace + stats, (2..) PMC passes (separate runs, no tracing domains
# besides --kernel-trace), (3) WRITE_SIZE / FETCH_SIZE calibration against a known-size memset.
# usage: tools/gpu_profile.sh <tag> <workload 256|512> <pipeline plain|fused>
# Output under gpurun_out/prof_<tag>/; copy summary.txt to profiles/ and feed it to tools/pmc_to_traffic.py.
TAG=${1:-run}
WL=${2:-256}
PIPE=${3:-fused}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# --pipeline: ONE kernel variant per kernel name (plain 32 B/voxel fill + tex0 march, or fused 36 B/voxel fill + distance-
# volume march), so that per-kernel averages belong to one variant; --no-batch: no 64-camera batch, 512^3 block or
# loopback probe
CMD="python bench.py --steps 10 --warmup 2 --workload $WL --pipeline $PIPE --no-cpu-baseline --no-batch --no-overlapped"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc --output-format csv -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_wr -o pmc --output-format csv -- $CMD > $OUT/pmc_wr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o pmc --output-format csv -- $CMD > $OUT/pmc_rd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/pmc_tcc -o pmc --output-format csv -- $CMD > $OUT/pmc_tcc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_calwr -o pmc --output-format csv -- python tools/pmc_calibrate.py $WL > $OUT/pmc_calwr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_calrd -o pmc --output-format csv -- python tools/pmc_calibrate.py $WL > $OUT/pmc_calrd.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $CMD"
  python tools/bench_line_of.py $OUT/trace.log
  cat $OUT/trace/trace_kernel_stats.csv
  echo
  echo "# per (kernel, grid size) of the same trace"
  python tools/kernel_trace_avg.py $OUT/trace 8
  echo
  echo "# PMC passes (one rocprofv3 --pmc run each; per-dispatch averages)"
  python tools/summarize_pmc.py $OUT
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
