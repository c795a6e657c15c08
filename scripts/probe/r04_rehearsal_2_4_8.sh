#!/bin/bash
# one rank's true share of the 2 / 4 / 8-rank strong split of the headline sphere (RCCL, the rank as
# the peer of its own exchanges), eager and with the step recorded, on the round's final kernels
mkdir -p gpurun_out/r4x
: > gpurun_out/r4x/rehearsal_2_4_8.jsonl
for size in 2 4 8; do
  for mode in "" "--step-graph"; do
    timeout -k 10 300 python scripts/measure_halo_exposure.py --scaling strong --size $size --steps 50 $mode > gpurun_out/r4x/reh_${size}${mode}.json 2> gpurun_out/r4x/reh_${size}${mode}.err || { tail -5 gpurun_out/r4x/reh_${size}${mode}.err; exit 1; }
    tail -1 gpurun_out/r4x/reh_${size}${mode}.json >> gpurun_out/r4x/rehearsal_2_4_8.jsonl
    python - "$size" "$mode" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4x/reh_%s%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("size", sys.argv[1], sys.argv[2] or "eager", "real", d.get("real_elements"), "ms/step", round(d.get("ms_per_step", 0), 3), "graph==eager", d.get("graph_equals_eager"))
PY
  done
done
