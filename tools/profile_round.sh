cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04final
(rocm-smi --showclocks --showuniqueid 2>&1 | grep -i "sclk\|mclk\|fclk\|unique") > gpurun_out/r04final/box_before.txt
python bench.py > gpurun_out/r04final/bench256.json 2> gpurun_out/r04final/bench256.err
python bench.py --workload 512 > gpurun_out/r04final/bench512.json 2> gpurun_out/r04final/bench512.err
for spec in "r04_256_ilv 256 fused_ilv" "r04_256_plain 256 plain" "r04_256_fused 256 fused" "r04_512_ilv 512 fused_ilv" "r04_512_plain 512 plain"; do
  bash tools/gpu_profile.sh $spec > /dev/null 2>&1
done
bash tools/gpu_profile_probe.sh r04_probe256 256 > /dev/null 2>&1
for t in r04_256_ilv r04_256_plain r04_256_fused r04_512_ilv r04_512_plain r04_probe256; do
  cp gpurun_out/prof_$t/summary.txt gpurun_out/r04final/${t}_rocprof_summary.txt
  cp gpurun_out/prof_$t/trace/trace_kernel_stats.csv gpurun_out/r04final/${t}_kernel_stats.csv 2>/dev/null
done
# the raw traces and counter CSVs (13 MB per run) stay on the box: gpurun copies back at most 64 MiB
for t in r04_256_ilv r04_256_plain r04_256_fused r04_512_ilv r04_512_plain r04_probe256; do
  rm -rf gpurun_out/prof_$t/trace gpurun_out/prof_$t/pmc_* gpurun_out/prof_$t/*.log
done
python tools/full_parity.py 256 512 1024 > gpurun_out/r04final/full_workload_parity.txt 2>&1
SDFV_BENCH_FORCE_MULTI=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 python bench.py --gpus 1 --no-cpu-baseline > gpurun_out/r04final/bench_rccl_loopback_256.json 2> gpurun_out/r04final/loopback.err
(rocm-smi --showclocks --showuniqueid 2>&1 | grep -i "sclk\|mclk\|fclk\|unique") > gpurun_out/r04final/box_after.txt
tail -3 gpurun_out/r04final/full_workload_parity.txt
