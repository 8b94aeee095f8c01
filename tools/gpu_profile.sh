#!/bin/bash
# Profiles `bench.py` on the GPU box: (1) kernel trace + stats, (2..) PMC passes (separate runs, no tracing
# domains besides --kernel-trace), (3) WRITE_SIZE / FETCH_SIZE calibration against a known-size memset.
# Output under gpurun_out/prof_<tag>/; copy the summaries to profiles/.
TAG=${1:-run}
WL=${2:-256}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# --no-tuned-placement: the placement probe launches the same kernel over the first slices only, which would mix
# shorter launches into the per-kernel averages (at 256^3 probe and step have the same shape, at 512^3 they do not)
CMD="python bench.py --steps 10 --warmup 2 --workload $WL --no-cpu-baseline --no-batch --no-tuned-placement"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc --output-format csv -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc_wr -o pmc --output-format csv -- $CMD > $OUT/pmc_wr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_rd -o pmc --output-format csv -- $CMD > $OUT/pmc_rd.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/pmc_tcc -o pmc --output-format csv -- $CMD > $OUT/pmc_tcc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_calwr -o pmc --output-format csv -- python tools/pmc_calibrate.py $WL > $OUT/pmc_calwr.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_calrd -o pmc --output-format csv -- python tools/pmc_calibrate.py $WL > $OUT/pmc_calrd.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats -- $CMD"
  cat $OUT/trace/trace_kernel_stats.csv
  echo
  echo "# PMC passes (one rocprofv3 --pmc run each; per-dispatch averages)"
  python tools/summarize_pmc.py $OUT
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
