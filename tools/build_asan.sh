#!/bin/bash
# Host-side AddressSanitizer builds (device code untouched: -fno-gpu-sanitize, plain gfx950 code objects) of the library and of the
# test oracle, for hunting host heap corruption under the GPU suite:
#   build/asan/libaero_stark.so   (AERO_LIB_PATH selects it)      build/asan/liboracle.so   (AERO_ORACLE_PATH selects it)
# Run with  LD_PRELOAD=$(tools/build_asan.sh --runtime)  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0  python3 -m pytest ...
set -e
# The runtime is GCC's libasan, not clang's: ROCm's compiler-rt intercepts hsa_amd_memory_pool_allocate (device ASan, needs xnack+,
# not available on this pool) and every hipMalloc then fails with "out of memory". clang's instrumentation calls only __asan_*
# entry points that libasan.so.6 exports, so the objects are linked WITHOUT a sanitizer runtime and resolve against the preloaded one.
# libstdc++ is preloaded with it (python does not link it; libasan's __cxa_throw interceptor needs it at start-up).
RT="/usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/x86_64-linux-gnu/libstdc++.so.6"
if [ "$1" = "--runtime" ]; then echo "$RT"; exit 0; fi
cd "$(dirname "$0")/.."
OUT=build/asan
mkdir -p $OUT
SRC=aero_amd/csrc
make -s -C $SRC gl_field_src.inc
SAN="-fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=()
for f in ntt hash stark air_kernels air_jit air_host prover capi capi_air verify comm_rccl comm_local export diag; do
  if [ ! -f $OUT/$f.o ] || [ $SRC/$f.hip -nt $OUT/$f.o ] || [ -n "$(find $SRC include -name '*.h*' -newer $OUT/$f.o 2>/dev/null | head -1)" ]; then
    hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -Wno-unused-result $SAN -c $SRC/$f.hip -o $OUT/$f.o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OUT/*.o -ldl -lhiprtc -o $OUT/libaero_stark.so
# the oracle with the SAME sanitizer runtime (g++ links libasan.so.6 dynamically)
g++ -O2 -g -march=x86-64-v3 -std=c++17 -fopenmp -fPIC -Wno-unused-function -fsanitize=address -fno-omit-frame-pointer -shared oracle/capi.cpp -o $OUT/liboracle.so
ls -la $OUT/*.so
