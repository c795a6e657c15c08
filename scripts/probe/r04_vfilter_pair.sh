#!/bin/bash
# the ocean models' two pre filters as a register-only line kernel (filters.h k_apply_vfilter_pair):
# parity (bitwise against CMDG_FUSED_COLUMNS=0), then the ocean bench with the pair off (1) and on (2)
mkdir -p gpurun_out/r4x
timeout -k 10 600 python -m pytest tests/test_gpu_ocean.py tests/test_gpu_split_explicit.py tests/test_gpu_split_explicit01.py tests/test_gpu_filters.py -x -q > gpurun_out/r4x/pytest_vfilter.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r4x/pytest_vfilter.log
[ $rc -ne 0 ] && exit $rc
for fz in 1 2 1 2; do CMDG_FUSED_COLUMNS=$fz python bench.py --workload ocean-split-explicit --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ocean fused_columns=$fz ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in d.get('kernels_ms', {}).items()})
"; done 2>&1 | tee gpurun_out/r4x/ab_ocean_vfilter.txt
python bench.py --workload ocean-split-explicit --steps 10 --warmup 3 > gpurun_out/r04_bench_ocean_48x48x16.json 2> gpurun_out/r4x/ocean.err; python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_ocean_48x48x16.json').read().strip().splitlines()[-1]); print('ocean', d['ms_per_step'], d['value'])
"
