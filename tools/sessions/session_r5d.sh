#!/bin/bash
# round 5 session d: counters of the two first-pass kernels (one wave per tile / 32 values per lane against two waves / 16 values)
AERO_NTT_F8X2=0 bash tools/ntt_gap.sh r5d_f8 20x2 20x72 > /dev/null 2>&1
AERO_NTT_F8X2=1 bash tools/ntt_gap.sh r5d_f8x2 20x2 20x72 > /dev/null 2>&1
for t in r5d_f8 r5d_f8x2; do echo "#### $t"; grep -A 30 "== ntt_fwd_first_pass_8" gpurun_out/$t/summary.txt | head -34; done
