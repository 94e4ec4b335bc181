#!/bin/bash
# The GPU suite against the host-ASan builds of tools/build_asan.sh (device code unchanged). Reports land in gpurun_out/asan/.
# usage: tools/run_asan_suite.sh [pytest args...]   (default: the whole -m gpu suite in the round-3 driver order)
OUT=$PWD/gpurun_out/asan
mkdir -p $OUT
[ -f build/asan/libaero_stark.so ] || tools/build_asan.sh > $OUT/build.log 2>&1      # build/ does not travel to the GPU box (.gpurunignore)
export AERO_LIB_PATH=$PWD/build/asan/libaero_stark.so AERO_ORACLE_PATH=$PWD/build/asan/liboracle.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0:log_path=$OUT/report:print_stats=0:handle_abort=1
export AERO_CRASH_TRACE=1 AERO_CRASH_LOG=$OUT/crash.log AERO_TEST_ORDER=${AERO_TEST_ORDER:-alpha}
RT="/usr/lib/x86_64-linux-gnu/libasan.so.6 /usr/lib/x86_64-linux-gnu/libstdc++.so.6"
t0=$(date +%s)
if [ $# -eq 0 ]; then set -- tests/ -q -m gpu -p no:cacheprovider; fi
LD_PRELOAD="$RT" python3 -m pytest "$@" > $OUT/pytest.log 2>&1
echo "rc=$? secs=$(( $(date +%s) - t0 ))" | tee $OUT/summary.txt
ls -la $OUT
tail -40 $OUT/pytest.log
for f in $OUT/report*; do [ -f "$f" ] && { echo "==== $f"; head -60 "$f"; }; done
