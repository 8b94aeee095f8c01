#!/bin/bash
# The round's evidence in one GPU call: default bench lines, one rocprofv3 summary per pipeline (each with the bench line of the
# very process that was traced and the build id / box it ran on), the driver's default command under the profiler, the
# progressive passes warm, PMC traffic, whole-workload parity, the N > 1 branch in RCCL loopback, this round's A/B runs.
# usage: tools/profile_round.sh <tag, e.g. r06>
R=${1:-r05}
cd $GRAFT_REPO_ROOT
D=gpurun_out/${R}final
mkdir -p $D
(rocm-smi --showclocks --showuniqueid 2>&1 | grep -i "sclk\|mclk\|fclk\|unique") > $D/box_before.txt
SDFV_BENCH_FULL_JSON=$D/bench256_full.json python bench.py > $D/bench256.json 2> $D/bench256.err
SDFV_BENCH_FULL_JSON=$D/bench512_full.json python bench.py --workload 512 > $D/bench512.json 2> $D/bench512.err
for spec in "${R}_256_ilv 256 fused_ilv" "${R}_256_plain 256 plain" "${R}_256_fused 256 fused" "${R}_512_ilv 512 fused_ilv" "${R}_512_plain 512 plain" "${R}_512_fused 512 fused"; do
  bash tools/gpu_profile.sh $spec > /dev/null 2>&1
done
bash tools/gpu_profile_probe.sh ${R}_probe256 256 > /dev/null 2>&1
for t in ${R}_256_ilv ${R}_256_plain ${R}_256_fused ${R}_512_ilv ${R}_512_plain ${R}_512_fused ${R}_probe256; do
  cp gpurun_out/prof_$t/summary.txt $D/${t}_rocprof_summary.txt
  cp gpurun_out/prof_$t/trace/trace_kernel_stats.csv $D/${t}_kernel_stats.csv 2>/dev/null
done
for s in 256 512; do
  bash tools/gpu_profile_pass.sh ${R}_pass$s $s > /dev/null 2>&1
  cp gpurun_out/prof_${R}_pass$s/summary.txt $D/${R}_pass${s}_rocprof_summary.txt
  cp gpurun_out/prof_${R}_pass$s/pass_traffic_$s.json $D/pass_traffic_$s.json
done
# the raw traces and counter CSVs (13 MB per run) stay on the box: gpurun copies back at most 64 MiB
for t in ${R}_256_ilv ${R}_256_plain ${R}_256_fused ${R}_512_ilv ${R}_512_plain ${R}_512_fused ${R}_probe256 ${R}_pass256 ${R}_pass512; do
  find gpurun_out/prof_$t -name "*.csv" -size +200k -delete 2>/dev/null
  rm -rf gpurun_out/prof_$t/trace gpurun_out/prof_$t/*/trace gpurun_out/prof_$t/*.log
done
python tools/full_parity.py 256 512 > $D/full_workload_parity.txt 2>&1
# round 6: the scan-load A/B (one process, one box), the no-op pass in its contexts, the read-stream microbenchmark, the ingest path
python tools/pass_loads_ab.py 256 512 > $D/${R}_pass_loads_ab.json 2> $D/pass_loads_ab.err
{ python tools/source_hash.py --stamp; python tools/noop_gap.py 512; python tools/noop_gap.py 256; } > $D/${R}_noop_gap.txt 2>&1
{ python tools/source_hash.py --stamp; ( cd tools/ubench && { [ -x ./read_stream ] || hipcc --offload-arch=gfx950 -O3 read_stream.hip -o read_stream; } && ./read_stream 512 && ./read_stream 64 && ./read_stream 2048 ); } > $D/${R}_read_stream_ubench.txt 2>&1
gcc -std=c11 -O2 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Iinclude tests/c/gyroid_provider.c -o /tmp/libgyroid_provider.so -lm
{ python tools/source_hash.py --stamp; for t in 1 4 16 32; do ./sdf-viewer_amd/sdf-viewer-host-bench --ingest /tmp/libgyroid_provider.so --side 384 --threads $t 2>&1 | grep -v Using; done; } > $D/${R}_ingest_bench.txt 2>&1
SDFV_BENCH_FORCE_MULTI=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 SDFV_BENCH_FULL_JSON=$D/bench_rccl_loopback_256_full.json \
  python bench.py --gpus 1 --no-cpu-baseline > $D/bench_rccl_loopback_256.json 2> $D/loopback.err
(rocm-smi --showclocks --showuniqueid 2>&1 | grep -i "sclk\|mclk\|fclk\|unique") > $D/box_after.txt
tail -3 $D/full_workload_parity.txt
