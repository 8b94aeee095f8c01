#!/bin/bash
# The host mirror (sdf-viewer_amd/host/) under the sanitizers ON THE GPU BOX: the ingest path's worker threads, pinned buffers
# and provider plumbing only run with a device.  Builds sanitizer variants of the host sources into /tmp (nothing in-tree),
# then
#   0. ThreadSanitizer: the WorkerPool's stress test (tests/c/worker_pool_stress.cpp);
#   1. ThreadSanitizer: sdf-viewer-host-bench --ingest (gyroid provider, per-point and batched) on 8 threads;
#   2. AddressSanitizer + UBSan: the same runs, and the 256^3 / 512^3 device-path loads;
#   3. AddressSanitizer + UBSan: tests/test_gpu_ingest.py + tests/test_gpu_host.py through a sanitizer build of the test library
#      (SDFV_HOST_TEST_LIB, LD_PRELOAD of the sanitizer runtimes under python).
# libamdhip64 / libsdfgrid are NOT instrumented: a report counts when one of its two accesses lies in our code (see the summary).
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
  g++ -O1 -g -std=c++17 -fsanitize=thread -I$H tests/c/worker_pool_stress.cpp -o $T/pool-tsan -pthread
  ls -la $T
} > $OUT/build.txt 2>&1

# (gcc 11's ThreadSanitizer runtime knows one layout of the address space: a kernel with 32 bits of mmap randomness puts the
# executable where it reports "unexpected memory mapping" -- its runs go with randomisation off where setarch may)
NOASLR=""
setarch x86_64 -R true 2>/dev/null && NOASLR="setarch x86_64 -R"
# 0. ThreadSanitizer over the WorkerPool alone (no device involved: tests/c/worker_pool_stress.cpp, 400 sessions of short runs)
for n in 6 16; do
  TSAN_OPTIONS="halt_on_error=0" timeout 600 $NOASLR $T/pool-tsan $n > $OUT/tsan_pool_$n.txt 2>&1
  echo "exit $?" >> $OUT/tsan_pool_$n.txt
done

# 1. ThreadSanitizer
for lib in libgyroid.so libgyroid_batch.so; do
  TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" timeout 900 $NOASLR $T/bench-tsan --ingest $T/$lib --side 96 --threads 8 \
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
# (with the sanitizer's dlopen in front, torch's own dlopen of libcaffe2_nvrtc.so no longer sees the caller's RUNPATH: name the
# directory)
TORCH_LIB=$(python -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))' 2>/dev/null)
ASAN_RT=$(g++ -print-file-name=libasan.so)
UBSAN_RT=$(g++ -print-file-name=libubsan.so)
LD_LIBRARY_PATH="$TORCH_LIB${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}" LD_PRELOAD="$ASAN_RT $UBSAN_RT" ASAN_OPTIONS="detect_leaks=0 halt_on_error=0" UBSAN_OPTIONS="print_stacktrace=1" \
  SDFV_HOST_TEST_LIB=$T/libsdfviewer_host_test_asan.so timeout 1500 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_host.py -x -q \
  -p no:cacheprovider > $OUT/asan_pytest.txt 2>&1
echo "exit $?" >> $OUT/asan_pytest.txt

# A report is OURS when one of its two ACCESSES has its first frame outside the sanitizer runtime in sdfviewer:: code or in the
# executable / test library; reports whose accesses both lie in libamdhip64 / libhsa-runtime64 (not instrumented: the runtime's
# own threads against the calling thread) are listed apart -- our frames appear in them only as the CALLER of a HIP entry point.
{
  head -1 $OUT/build.txt
  echo "address-space randomisation off for the ThreadSanitizer runs: ${NOASLR:-no (setarch refused)}"
  python - $OUT/tsan_*.txt $OUT/asan_*.txt <<'PY'
import os, re, sys
for path in sys.argv[1:]:
    txt = open(path, errors="replace").read()
    reports = re.split(r"(?=WARNING: ThreadSanitizer|ERROR: AddressSanitizer)", txt)[1:]
    ubsan = len(re.findall(r"runtime error:", txt))
    ours = 0
    for r in reports:
        for block in re.findall(r"^  (?:Previous )?(?:[Aa]tomic )?(?:[Ww]rite|[Rr]ead) of size.*?\n((?:    #.*\n)+)", r, re.M):
            frames = [f for f in block.splitlines() if "libtsan" not in f and "libasan" not in f]
            if frames and re.search(r"sdfviewer::|bench-[at]san|pool-tsan|libsdfviewer_host_test_asan|libgyroid", frames[0]):
                ours += 1
                break
        else:
            if r.startswith("ERROR: AddressSanitizer") and re.search(r"#[0-3] .*(sdfviewer::|bench-asan|libsdfviewer_host_test_asan)", r):
                ours += 1
    exit_line = ([l for l in txt.splitlines() if l.startswith("exit ")] or ["exit ?"])[-1]
    results = sum(1 for l in txt.splitlines() if l.startswith("{") or l.startswith("ok "))
    print(f"{os.path.basename(path)}: reports {len(reports)} (+ {ubsan} UBSan), with an access in our code {ours}, "
          f"inside the HIP / HSA runtime only {len(reports) - ours}, {exit_line}, {results} result line(s)")
PY
  tail -3 $OUT/asan_pytest.txt
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
