mkdir -p gpurun_out/r4z
for fz in 0 1 2 3 4 0 1 2 3 4; do CMDG_FUSED_COLUMNS=$fz python bench.py --workload ocean-split-explicit --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ocean fused level $fz ms/step %.3f' % d['ms_per_step'], {k: round(v['avg_ms']*1e3,1) for k, v in d.get('kernels_ms', {}).items()})
"; done 2>&1 | tee gpurun_out/r4z/ab_ocean_fused_levels.txt
