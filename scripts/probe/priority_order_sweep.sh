#!/bin/bash
# Localises the stream-priority ordering failure of the 2-rank split-explicit test: the test is run
# once per CMDG_DBG_SYNC bit (engine.h) with CMDG_HALO_PRIORITY=1; a bit under which it passes names
# the class of event edges that does not hold.  usage: scripts/probe/priority_order_sweep.sh [outdir] [masks...]
OUT=${1:-gpurun_out/prio_sweep}; shift
mkdir -p "$OUT"
: > "$OUT/summary.txt"
MASKS=${@:-"off 0 1 2 4 8 16 32 64 128 256 512"}
T='tests/test_gpu_split_explicit.py::test_partitioned_split_explicit_matches_single_rank'
for m in $MASKS; do
  if [ "$m" = "off" ]; then pr=0; mm=0; else pr=1; mm=$m; fi
  CMDG_HALO_PRIORITY=$pr CMDG_DBG_SYNC=$mm timeout -k 10 300 python -m pytest "$T" -x -q -k "2" > "$OUT/mask_$m.log" 2>&1
  rc=$?
  echo "priority=$pr mask=$mm rc=$rc $(tail -n 1 $OUT/mask_$m.log)" | tee -a "$OUT/summary.txt"
done
exit 0
