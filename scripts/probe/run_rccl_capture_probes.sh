#!/bin/bash
# Runs every mode of rccl_capture_probe.py on both stacks, one fresh process each, under a timeout;
# writes one line per run to $OUT/summary.txt and the process output next to it.
# usage: scripts/probe/run_rccl_capture_probes.sh [outdir]
OUT=${1:-gpurun_out/rccl_capture}
mkdir -p "$OUT"
: > "$OUT/summary.txt"
for stack in torch system; do
  for mode in eager origin-global origin-threadlocal origin-relaxed fork-memcpy fork-relaxed fork-global two-groups-origin send-recv-split; do
    log="$OUT/${stack}_${mode}.log"
    timeout -k 5 120 python scripts/probe/rccl_capture_probe.py $mode $stack > "$log" 2>&1
    rc=$?
    last=$(grep -v '^$' "$log" | tail -n 1 | cut -c1-160)
    echo "$stack $mode rc=$rc | $last" | tee -a "$OUT/summary.txt"
    # a timeout (124/137) means a hung replay: stop probing, the GPU may be wedged
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" | tee -a "$OUT/summary.txt"; exit 0; fi
  done
done
exit 0
