#!/bin/bash
# Round evidence that tools/profile_round.sh does not cover: the 64-camera batch under the profiler (kernel trace + VALU / VMEM
# counters -> profiles/raymarch_batch_valu.json) at 256^3 and 512^3, and the point / mesher front-end kernels.  usage: <tag>
R=${1:-r05}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
D=gpurun_out/${R}batch; mkdir -p $D
for S in 256 512; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $D/trace$S -o trace --output-format csv -- python tools/batch_valu.py run $S > $D/trace$S.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- python tools/batch_valu.py run $S   (fused fill, y-pair / y-interleaved volume, 3 batches of 64 cameras)"; cat $D/trace$S/trace_kernel_stats.csv; } > $D/${R}_batch64_${S}_rocprof_summary.txt
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $D/pmc$S -o pmc --output-format csv -- python tools/batch_valu.py run $S > $D/pmc$S.log 2>&1
  python tools/batch_valu.py reduce $D/pmc$S $S > $D/batch_valu_$S.json 2> $D/reduce$S.err
  rm -rf $D/trace$S $D/pmc$S
done
cp profiles/raymarch_batch_valu.json $D/raymarch_batch_valu.json
python tools/points_bench.py > $D/${R}_points_bench.json 2> $D/points.err
cat $D/batch_valu_256.json $D/batch_valu_512.json; tail -c 600 $D/${R}_points_bench.json
