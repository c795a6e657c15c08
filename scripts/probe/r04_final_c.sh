# round-4 closing pass, part C: bench lines retaken with the PMC profiles of this tree in place
mkdir -p gpurun_out/r4y
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_heldsuarez_n30_bench.json 2> gpurun_out/r4y/hs.err; python -c "
import json
d=json.loads(open('gpurun_out/r04_heldsuarez_n30_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('hs', d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['traffic_over_needed'], d['cpu_baseline']['value'], d['cpu_baseline']['threads'])
"
python bench.py --workload risingbubble --steps 20 --warmup 5 > gpurun_out/r04_risingbubble_8000_bench.json 2> gpurun_out/r4y/rb.err; python -c "
import json
d=json.loads(open('gpurun_out/r04_risingbubble_8000_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('rb', d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['traffic_over_needed'])
"
python bench.py --workload bomex --steps 10 --warmup 3 > gpurun_out/r04_bomex_n6_8192_bench.json 2> gpurun_out/r4y/bomex.err; python -c "
import json
d=json.loads(open('gpurun_out/r04_bomex_n6_8192_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bomex', d['ms_per_step'], d['value'], r['frac'], r['traffic'], r['traffic_over_needed'])
"
BENCH_REHEARSE_NRANK=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu > gpurun_out/r4y/rehearse_n1.json 2> gpurun_out/r4y/rehearse_n1.err; echo rehearse rc=$?; python -c "
import json
d=json.loads(open('gpurun_out/r4y/rehearse_n1.json').read().strip().splitlines()[-1]); print('rehearsal', d['n_gpus'], d['scaling'], d['ms_per_step'], list(d.keys())[:30]); print(d.get('halo'))
"
for fl in "" "--step-graph" "--step-graph --async-run"; do python scripts/measure_halo_exposure.py --scaling strong --size 8 --steps 100 $fl 2> /dev/null | tail -1 > "gpurun_out/r4y/exposure_strong$(echo $fl | tr -d ' ').json"; python -c "
import sys, json
d=json.loads(open('gpurun_out/r4y/exposure_strong$(echo $fl | tr -d ' ').json').read()); print('exposure [$fl]', {k:(round(d[k],3) if isinstance(d[k],float) else d[k]) for k in ('ms_per_step','host_enqueue_ms_per_step','graph_steps_replayed','graph_equals_eager','async_run') if k in d})
"; done
python scripts/measure_halo_exposure.py --scaling weak --size 8 --steps 30 2>/dev/null | tail -1 > gpurun_out/r4y/exposure_weak.json; python -c "
import json
d=json.loads(open('gpurun_out/r4y/exposure_weak.json').read()); print('weak eager', round(d['ms_per_step'],3), round(d['host_enqueue_ms_per_step'],3))
"
python scripts/measure_halo_exposure.py --scaling weak --size 8 --steps 30 --step-graph 2>/dev/null | tail -1 > gpurun_out/r4y/exposure_weak_graph.json; python -c "
import json
d=json.loads(open('gpurun_out/r4y/exposure_weak_graph.json').read()); print('weak graph', round(d['ms_per_step'],3), round(d['host_enqueue_ms_per_step'],3), d.get('graph_equals_eager'))
"
