#!/bin/bash
# round 5 session b: hand-written carry chains (GL_ASM) against the compiler's, microbenchmarks only
mkdir -p gpurun_out/r5b
for v in asm base; do
  echo "== field $v"; timeout 40 tools/ubench_field_$v
  echo "== dft $v"; timeout 40 tools/ubench_dft_$v
done 2>&1 | tee gpurun_out/r5b/ubench.txt
