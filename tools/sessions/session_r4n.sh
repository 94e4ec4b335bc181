#!/bin/bash
# Round 4, session n: interpolation reads the caller's matrix (no device-to-device copy), 512-thread LDS passes on small launches by default.
OUT=gpurun_out/r4n; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_full_configs.py tests/test_gpu_c_abi_host.py -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/tests.txt
for i in 1 2; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d.get('single_proof_ms'), d.get('single_proof_ms_hbm_resident'))" | tee -a $OUT/bench.txt
done
python3 tools/ntt_ab.py 20x1 20x2 21x1 19x2 16x2 2>&1 | tail -1 | tee -a $OUT/bench.txt
