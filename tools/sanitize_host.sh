#!/bin/bash
# The host mirror (sdf-viewer_amd/host/) under the sanitizers ON THE GPU BOX: the ingest path's worker threads, pinned buffers
# and provider plumbing only run with a device.  Builds sanitizer variants of the host sources into /tmp (nothing in-tree),
# then
#   1. ThreadSanitizer: sdf-viewer-host-bench --ingest (gyroid provider, per-point and batched) on 8 threads;
#   2. AddressSanitizer + UBSan: the same runs, and the 256^3 / 512^3 device-path loads;
#   3. AddressSanitizer + UBSan: tests/test_gpu_ingest.py + tests/test_gpu_host.py through a sanitizer build of the test library
#      (SDFV_HOST_TEST_LIB, LD_PRELOAD of the sanitizer runtimes under python).
# libamdhip64 / libsdfgrid are NOT instrumented: a report counts when one of its frames lies in sdfviewer:: code.
# usage (gpurun): timeout 2400 bash tools/sanitize_host.sh ; summary in gpurun_out/sanitize/summary.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 2
OUT=gpurun_out/sanitize
mkdir -p $OUT
H=sdf-viewer_amd/host
CORE="$H/sdf_demo.cpp $H/sdf_viewer.cpp $H/sdf_viewer_ingest.cpp $H/provider_sdf.cpp $H/scene.cpp $H/mesh.cpp"
FLAGS="-O1 -g -std=c++17 -ffp-contract=off -fPIC -fno-omit-frame-pointer -Wno-unused-function -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
LINK="-Lsdf-viewer_amd -lsdfgrid -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$PWD/sdf-viewer_amd -ldl -pthread"
T=/tmp/sdfv_sanitize
mkdir -p $T
{
  python tools/source_hash.py --stamp
  gcc -std=c11 -O2 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Iinclude tests/c/gyroid_provider.c -o $T/libgyroid.so -lm
  gcc -std=c11 -O2 -ffp-contract=off -fPIC -shared -fvisibility=hidden -DGYROID_BATCH -Iinclude tests/c/gyroid_provider.c -o $T/libgyroid_batch.so -lm
  g++ $FLAGS -fsanitize=thread -o $T/bench-tsan $H/host_load_bench.cpp $CORE $LINK
  g++ $FLAGS -fsanitize=address,undefined -o $T/bench-asan $H/host_load_bench.cpp $CORE $LINK
  g++ $FLAGS -fsanitize=address,undefined -fvisibility=hidden -shared -o $T/libsdfviewer_host_test_asan.so $CORE $H/host_capi.cpp $LINK
  ls -la $T
} > $OUT/build.txt 2>&1

# 1. ThreadSanitizer
for lib in libgyroid.so libgyroid_batch.so; do
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" timeout 900 $T/bench-tsan --ingest $T/$lib --side 96 --threads 8 \
    > $OUT/tsan_$lib.txt 2>&1
  echo "exit $?" >> $OUT/tsan_$lib.txt
done
# 2. ASan + UBSan, executable
for lib in libgyroid.so libgyroid_batch.so; do
  ASAN_OPTIONS="detect_leaks=0 halt_on_error=0" UBSAN_OPTIONS="print_stacktrace=1" timeout 900 $T/bench-asan --ingest $T/$lib --side 128 --threads 8 \
    > $OUT/asan_$lib.txt 2>&1
  echo "exit $?" >> $OUT/asan_$lib.txt
done
for side in 256 512; do
  ASAN_OPTIONS="detect_leaks=0 halt_on_error=0" UBSAN_OPTIONS="print_stacktrace=1" timeout 900 $T/bench-asan --side $side > $OUT/asan_load_$side.txt 2>&1
  echo "exit $?" >> $OUT/asan_load_$side.txt
done
# 3. ASan + UBSan under pytest
ASAN_RT=$(g++ -print-file-name=libasan.so)
UBSAN_RT=$(g++ -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN_RT $UBSAN_RT" ASAN_OPTIONS="detect_leaks=0 halt_on_error=0" UBSAN_OPTIONS="print_stacktrace=1" \
  SDFV_HOST_TEST_LIB=$T/libsdfviewer_host_test_asan.so timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_host.py -x -q \
  -p no:cacheprovider > $OUT/asan_pytest.txt 2>&1
echo "exit $?" >> $OUT/asan_pytest.txt

{
  head -1 $OUT/build.txt
  for f in $OUT/tsan_*.txt $OUT/asan_*.txt; do
    reports=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:" $f)
    ours=$(grep -c "sdfviewer::" $f)
    echo "$(basename $f): sanitizer reports $reports, lines naming sdfviewer:: $ours, $(grep '^exit' $f | tail -1), $(grep -c '^{' $f) result line(s)"
  done
  tail -3 $OUT/asan_pytest.txt
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
