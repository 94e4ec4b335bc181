#!/bin/bash
# Build the library's HOST code against the stand-in runtime with a sanitizer and run the host-logic exercise (no GPU needed).
# usage: tools/hipstub/run.sh thread|address
set -e
SAN=${1:-thread}
cd "$(dirname "$0")/../.."
OUT=build/hipstub_$SAN
mkdir -p $OUT
SRC=aero_amd/csrc
make -s -C $SRC gl_field_src.inc
CLANG=/opt/rocm/lib/llvm/bin/clang++
pids=()
for f in ntt hash stark air_kernels air_jit air_host prover capi capi_air verify comm_rccl comm_local export diag numa; do
  if [ ! -f $OUT/$f.o ] || [ $SRC/$f.hip -nt $OUT/$f.o ] || [ -n "$(find $SRC include -name '*.h*' -newer $OUT/$f.o 2>/dev/null | head -1)" ]; then
    # host side only: the kernels' device code is not needed (they never run here)
    hipcc --offload-host-only --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-result -fsanitize=$SAN -fno-omit-frame-pointer -Wno-option-ignored -c $SRC/$f.hip -o $OUT/$f.o &
    pids+=($!)
    if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
  fi
done
wait
$CLANG -O1 -g -std=c++17 -fsanitize=$SAN -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tools/hipstub/hipstub.cpp tools/hipstub/host_logic.cpp $OUT/*.o \
   -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib -Wl,--unresolved-symbols=ignore-all -lpthread -ldl -o $OUT/host_logic   # the kernels' fat binaries are absent on purpose (host-only objects)
if [ "$SAN" = thread ]; then export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 exitcode=66"; else export ASAN_OPTIONS="detect_leaks=1"; fi
$OUT/host_logic
