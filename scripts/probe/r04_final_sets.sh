#!/bin/bash
# round-4 closing pass on the final kernel sources: the three profile sets (bench line, kernel stats, PMC),
# then the bench lines once more with the PMC files in place (so that `traffic` carries this tree's digest)
mkdir -p gpurun_out/r4x
bash scripts/profile_bench.sh r04_heldsuarez_n30 "--steps 20 --warmup 5" 2>&1 | tail -3
bash scripts/profile_bench.sh r04_risingbubble_8000 "--workload risingbubble --steps 20 --warmup 5" 2>&1 | tail -3
bash scripts/profile_bench.sh r04_bomex_n6_8192 "--workload bomex --steps 10 --warmup 3" 2>&1 | tail -3
cp gpurun_out/r04_heldsuarez_n30_pmc_hbm_per_launch.json gpurun_out/r04_risingbubble_8000_pmc_hbm_per_launch.json gpurun_out/r04_bomex_n6_8192_pmc_hbm_per_launch.json profiles/
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_heldsuarez_n30_bench.json 2> gpurun_out/r4x/hs.err
python bench.py --workload risingbubble --steps 20 --warmup 5 > gpurun_out/r04_risingbubble_8000_bench.json 2> gpurun_out/r4x/rb.err
python bench.py --workload bomex --steps 10 --warmup 3 > gpurun_out/r04_bomex_n6_8192_bench.json 2> gpurun_out/r4x/bomex.err
python - <<'PY'
import json
for t in ("r04_heldsuarez_n30", "r04_risingbubble_8000", "r04_bomex_n6_8192"):
    d = json.loads(open("gpurun_out/%s_bench.json" % t).read().strip().splitlines()[-1]); r = d["roofline"]
    print(t, round(d["ms_per_step"], 3), "%.3e" % d["value"], round(r["frac"], 3), r["traffic"], r["traffic_over_needed"])
PY
