#!/bin/bash
# Host-side AddressSanitizer builds (device code untouched: -fno-gpu-sanitize, plain gfx950 code objects) of the library and of the
# test oracle, for hunting host heap corruption under the GPU suite:
#   build/asan/libaero_stark.so   (AERO_LIB_PATH selects it)      build/asan/liboracle.so   (AERO_ORACLE_PATH selects it)
# Run with  LD_PRELOAD=$(tools/build_asan.sh --runtime)  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0  python3 -m pytest ...
set -e
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
if [ "$1" = "--runtime" ]; then echo $RT; exit 0; fi
cd "$(dirname "$0")/.."
OUT=build/asan
mkdir -p $OUT
SRC=aero_amd/csrc
make -s -C $SRC gl_field_src.inc
SAN="-fsanitize=address -fno-gpu-sanitize -shared-libsan -fno-omit-frame-pointer"
pids=()
for f in ntt hash stark air_kernels air_jit air_host prover capi capi_air verify comm_rccl comm_local export diag; do
  if [ ! -f $OUT/$f.o ] || [ $SRC/$f.hip -nt $OUT/$f.o ] || [ -n "$(find $SRC include -name '*.h*' -newer $OUT/$f.o 2>/dev/null | head -1)" ]; then
    hipcc --offload-arch=gfx950 -O2 -g -std=c++17 -fPIC -Wno-unused-result $SAN -c $SRC/$f.hip -o $OUT/$f.o &
    pids+=($!)
    if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread $SAN $OUT/*.o -ldl -lhiprtc -o $OUT/libaero_stark.so
# the oracle with the SAME sanitizer runtime (clang's; gcc's libasan cannot share a process with it)
/opt/rocm/lib/llvm/bin/clang++ -O2 -g -march=x86-64-v3 -std=c++17 -fopenmp -fPIC -Wno-unused-function -fsanitize=address -shared-libsan -fno-omit-frame-pointer \
    -shared oracle/capi.cpp -o $OUT/liboracle.so
ls -la $OUT/*.so
