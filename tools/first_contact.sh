#!/bin/bash
# First contact with a multi-GPU node: the N > 1 path has only ever run over gloo and over RCCL in loopback.
# This runs, in order and each under the bench's own watchdog (which names the collective stage a rank is stuck in and still
# prints the measured line), the steps a first run should take -- smallest first, each writing ONE short line to
# gpurun_out/first_contact/<step>.json (stderr beside it) -- and stops at the first step that fails.
#   usage: tools/first_contact.sh [N=8]        (from the repository root, on a node with N GPUs)
#          SDFV_BENCH_BACKEND=gloo tools/first_contact.sh 2    rehearsal on ONE GPU (ranks share it, gloo carries the exchanges)
# Steps:  1  --gpus 2, tiny workload: ncclCommInitRank sees two devices; fill step + ghosts + sharded march verified; RCCL's
#            warnings / topology (NCCL_DEBUG=INFO, INIT + GRAPH) and rocm-smi's link types summarised in 1_summary.json
#         2  --gpus 2, the default workload (256^3 + 1080p): the first real number
#         3  --gpus N, 256^3 per rank, slab geometry: what the driver's SCALE run does at N
#         4  --gpus N, config 4: cube geometry at 512^3 per rank (N = 8: 1024^3), halo 33.5 MB per direction
#         5  --gpus N, config 5: the 64-camera batch, all three splits with gather_ms
#         6  the sharded march by itself (sdfv_slab_march over the library communicator), 1080p, self-checked
N=${1:-8}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/first_contact
mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8
export SDFV_BENCH_WATCHDOG_S=${SDFV_BENCH_WATCHDOG_S:-600} SDFV_BENCH_EXTRAS_S=${SDFV_BENCH_EXTRAS_S:-240}
PORT=29810
run() {  # name, ranks, bench arguments...
    local name=$1 ranks=$2; shift 2
    PORT=$((PORT + 1))
    echo "[first_contact] $name: $ranks ranks: bench.py $*" >&2
    SDFV_BENCH_FULL_JSON=$OUT/${name}_full.json timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks \
        --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $ranks --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
    local rc=$?
    tail -n 1 $OUT/$name.json
    if [ $rc -ne 0 ] || ! tail -n 1 $OUT/$name.json | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); ok = d.get("sharded_fill_verified") is True and not d.get("incomplete"); sys.exit(0 if ok else 1)'; then
        echo "[first_contact] $name FAILED (rc $rc): see $OUT/$name.err (the watchdog's stage report and stacks)" >&2
        grep -a "WATCHDOG\|library communicator unavailable\|Error\|error" $OUT/$name.err | tail -n 12 >&2
        exit 1
    fi
}
# Step 1 explains itself should it be slow or fail: RCCL's own warnings and what it says about the topology go into the step's
# stderr file, and a summary line -- the ranks RCCL connected, its warnings, the links it found, rocm-smi's link-type matrix --
# is printed and kept in $OUT/1_summary.json.
NCCL_DEBUG=${NCCL_DEBUG:-INFO} NCCL_DEBUG_SUBSYS=${NCCL_DEBUG_SUBSYS:-INIT,GRAPH,ENV} \
run 1_two_ranks_tiny 2 --steps 3 --warmup 1 --workload 64 --config4-side 32 --prewarm-ms 5 --per-step-samples 4
python - "$OUT" <<'PY' | tee $OUT/1_summary.json
import json, re, subprocess, sys
out = sys.argv[1]
line = json.loads(open(f"{out}/1_two_ranks_tiny.json").read().strip().splitlines()[-1])
err = open(f"{out}/1_two_ranks_tiny.err", errors="replace").read().splitlines()
warn = [l.strip()[:200] for l in err if " NCCL WARN " in l or "RCCL WARN" in l]
topo = [l.strip()[:200] for l in err if re.search(r"XGMI|xGMI|P2P|Channel \d+/\d+ :|nChannels|via ", l)]
try:
    smi = subprocess.run(["rocm-smi", "--showtopotype"], capture_output=True, text=True, timeout=60).stdout
    xgmi_per_gpu = [row.count("XGMI") for row in smi.splitlines() if row.startswith("GPU")]
except Exception as e:  # noqa: BLE001
    smi, xgmi_per_gpu = f"{type(e).__name__}: {e}", None
print(json.dumps({"step": "1_two_ranks_tiny", "rccl_ranks": line.get("rccl_ranks"), "torch_world_size": line.get("torch_world_size"),
                  "backend": line.get("backend"), "halo_transport": line.get("halo_transport"), "ms_per_step_fill": line.get("ms_per_step_fill"),
                  "nccl_warnings": warn[:20], "rccl_topology_lines": len(topo), "rccl_topology_sample": topo[:12],
                  "rocm_smi_xgmi_peers_per_gpu": xgmi_per_gpu}))
PY
run 2_two_ranks_256 2 --steps 20 --warmup 5
run 3_n_ranks_256 $N --steps 20 --warmup 5 --no-config4 --no-batch
run 4_n_ranks_config4 $N --steps 10 --warmup 3 --workload 512 --weak-geometry cube --no-batch --no-config4
run 5_n_ranks_config5 $N --steps 10 --warmup 3 --no-config4 --batch-split all
if [ "$SDFV_BENCH_BACKEND" = "gloo" ]; then  # (a rehearsal on one GPU: ranks share the device over gloo; the library communicator needs RCCL)
    echo "[first_contact] rehearsal over gloo: step 6 (sdfv_slab_march over RCCL) skipped; steps 1-5 passed; lines in $OUT/" >&2
    exit 0
fi
PORT=$((PORT + 1))
echo "[first_contact] 6_sharded_march: $N ranks: tools/sharded_march_ranks.py" >&2
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
    tools/sharded_march_ranks.py > $OUT/6_sharded_march.json 2> $OUT/6_sharded_march.err || { echo "[first_contact] 6_sharded_march FAILED" >&2; tail -n 12 $OUT/6_sharded_march.err >&2; exit 1; }
tail -n 1 $OUT/6_sharded_march.json
echo "[first_contact] all steps passed; lines in $OUT/" >&2
