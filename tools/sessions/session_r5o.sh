#!/bin/bash
# round 5 session o: completion word polled at the tree roots - parity, then one proof alone with and without polling; headline bench
R=$PWD; O=$R/gpurun_out/r5o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_aux.py tests/test_gpu_random_configs.py tests/test_gpu_sharded_local.py -x -q -m gpu 2>&1 | tail -4 | tee $O/parity.txt
for i in 1 2 3; do python3 tools/single_latency.py 20 2 300; AERO_POLL_FLAGS=0 python3 tools/single_latency.py 20 2 300; done | tee $O/single.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-air-program 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d.get('single_proof_ms'), d.get('single_proof_ms_hbm_resident'), d['pcie'], d['host_placement'])" | tee -a $O/single.txt
