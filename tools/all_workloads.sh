#!/bin/bash
# Headline numbers of every bench workload (no CPU baseline). usage: bash tools/all_workloads.sh
for w in "fib_2^20x2_blowup8_blake2s_base" "fib_2^20x2_blowup8_blake2s_quadratic" "fib_2^24x2_blowup8_blake2s_base" "fib_2^20x72_blowup8_blake2s_base" "standin_miden_shape_2^22x(72+9aux)_deg8_fold4"; do
  python bench.py --workload "$w" --no-cpu-baseline --no-air-program --steps 12 --warmup 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-52s %7.1f M cells/s (H2D incl.; %7.1f HBM-resident)  in flight %d  single proof %8.3f ms (%8.3f resident)  dominant %s (%.1f %% of HBM peak alone on the GPU)  %.2f GB/proof' % (d['config']['workload'], d['value']/1e6, d['hbm_resident_value']/1e6, d['config']['proofs_in_flight_per_gpu'], d['single_proof_ms'], d['single_proof_ms_hbm_resident'], d['roofline']['kernel'], 100*d['roofline']['one_proof_in_flight']['frac'], d['device_bytes_peak_per_proof_in_flight']/1e9))"
done
