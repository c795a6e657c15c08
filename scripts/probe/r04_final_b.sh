# round-4 closing pass, part B: ocean fused-column A/B + tests, the other workloads' profile sets and bench lines
mkdir -p gpurun_out/r4y
timeout -k 10 600 python -m pytest tests/test_gpu_ocean.py tests/test_gpu_split_explicit.py tests/test_gpu_split_explicit01.py tests/test_gpu_integrals.py -x -q > gpurun_out/r4y/pytest_ocean.log 2>&1; echo pytest ocean rc=$?; tail -3 gpurun_out/r4y/pytest_ocean.log
for fz in 0 1 0 1; do CMDG_FUSED_COLUMNS=$fz python bench.py --workload ocean-split-explicit --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ocean fused_columns=$fz ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in d.get('kernels_ms', {}).items()})
"; done 2>&1 | tee gpurun_out/r4y/ab_ocean_fused_columns.txt
bash scripts/profile_bench.sh r04_risingbubble_8000 "--workload risingbubble --steps 20 --warmup 5" 2>&1 | tail -4
bash scripts/profile_bench.sh r04_bomex_n6_8192 "--workload bomex --steps 10 --warmup 3" 2>&1 | tail -4
python bench.py --workload bomex --bomex-ne 32 --steps 5 --warmup 2 --no-cpu > gpurun_out/r04_bench_bomex_n6_65536.json 2> gpurun_out/r4y/bomex65536.err; tail -c 300 gpurun_out/r04_bench_bomex_n6_65536.json
python bench.py --workload ocean-split-explicit --steps 10 --warmup 3 > gpurun_out/r04_bench_ocean_48x48x16.json 2> gpurun_out/r4y/ocean.err; python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_ocean_48x48x16.json').read().strip().splitlines()[-1]); print('ocean', d['ms_per_step'], d.get('cpu_baseline'))
"
python bench.py --workload advdiff-brick --steps 20 --warmup 5 --no-cpu > gpurun_out/r04_bench_advdiff_ne32.json 2> gpurun_out/r4y/advdiff.err; python -c "
import json
d=json.loads(open('gpurun_out/r04_bench_advdiff_ne32.json').read().strip().splitlines()[-1]); print('advdiff', d['ms_per_step'], d['value'])
"
