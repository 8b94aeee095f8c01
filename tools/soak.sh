#!/bin/bash
# Randomised parity soak on the GPU box: the seeded sweeps of the test suite re-run with fresh seeds and more trials
# until the time budget (seconds, default 300) is spent.  Any mismatch stops the loop and prints the failing seed.
BUDGET=${1:-300}
cd $GRAFT_REPO_ROOT
END=$(( $(date +%s) + BUDGET ))
SEED=${2:-1000}
N=0
while [ $(date +%s) -lt $END ]; do
  SEED=$((SEED + 1))
  if ! SDFV_SOAK_SEED=$SEED SDFV_SOAK_TRIALS=40 timeout 900 python -m pytest -x -q \
        tests/test_gpu_fill.py::test_randomised_parameters_and_grids \
        tests/test_gpu_raymarch.py::test_randomised_cameras_grids_and_boxes \
        tests/test_gpu_raymarch.py::test_randomised_sweep_of_the_hand_written_march_loop \
        tests/test_gpu_raymarch.py::test_randomised_tile_orders \
        tests/test_gpu_points.py::test_random_points_match_oracle \
        tests/test_gpu_mesh_extract.py::test_randomised_extractions_match_numpy_restatement \
        tests/test_gpu_sharded_march.py::test_randomised_slabs_and_cameras \
        tests/test_gpu_ingest.py::test_randomised_ingest_loads_and_edits > gpurun_out/soak_last.log 2>&1; then
    echo "SOAK FAILURE at seed $SEED"; grep -v "^RCCL\|^HIP \|^ROCm\|^Host\|^Libr" gpurun_out/soak_last.log | tail -40
    exit 1
  fi
  N=$((N + 1))
done
echo "soak: $N rounds of 8 sweeps x 40 trials passed, seeds $(( SEED - N + 1 ))..$SEED"
