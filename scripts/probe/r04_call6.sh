mkdir -p gpurun_out/r4e
CMDG_HALO_PRIORITY=1 CMDG_OCEAN_FAST_PRIORITY=0 CMDG_DBG_WORK_MEMSET=null python scripts/probe/priority_order_diag.py > gpurun_out/r4e/diag_legacy_memset.txt 2>&1; echo "legacy fill, fast model at default priority:"; grep worst gpurun_out/r4e/diag_legacy_memset.txt
CMDG_HALO_PRIORITY=1 CMDG_OCEAN_FAST_PRIORITY=0 python scripts/probe/priority_order_diag.py > gpurun_out/r4e/diag_fixed.txt 2>&1; echo "fixed, same setting:"; grep worst gpurun_out/r4e/diag_fixed.txt
python scripts/measure_halo_exposure.py --scaling strong --size 8 --steps 100 --step-graph 2> gpurun_out/r4e/exp_graph.err | tail -1 > gpurun_out/r4e/exposure_strong_graph.json
python -c "
import json
d=json.load(open('gpurun_out/r4e/exposure_strong_graph.json')); print('graph', {k:d[k] for k in ('ms_per_step','host_enqueue_ms_per_step','graph_steps_replayed','graph_equals_eager','graph_check_nonfinite_values','graph_check_steps_replayed') if k in d})
"
bash scripts/ab_env.sh "--steps 20 --warmup 5 --no-parity --no-secondary" CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 2>&1 | tee gpurun_out/r4e/ab_pairs_hs.txt
bash scripts/ab_env.sh "--workload risingbubble --steps 20 --warmup 5 --no-parity --no-secondary" CMDG_TENDENCY_PAIRS=0 CMDG_TENDENCY_PAIRS=1 2>&1 | tee gpurun_out/r4e/ab_pairs_rb.txt
for p in 0 1 0 1; do CMDG_OCEAN_FAST_PRIORITY=$p python bench.py --workload ocean-split-explicit --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ocean fast_priority=$p ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in d.get('kernels_ms', {}).items()})
"; done 2>&1 | tee gpurun_out/r4e/ab_ocean_priority.txt
CMDG_TENDENCY_PAIRS=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_halo_direct.py -x -q > gpurun_out/r4e/pytest_pairs.log 2>&1; echo pytest pairs rc=$?; tail -3 gpurun_out/r4e/pytest_pairs.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4e/bench_default.json 2> gpurun_out/r4e/bench_default.err; python -c "
import json
d=json.loads(open('gpurun_out/r4e/bench_default.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); print(json.dumps(d['cpu_baseline'])[:1500])
"
