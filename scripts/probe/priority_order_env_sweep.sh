#!/bin/bash
# Second localisation pass for the priority ordering failure: scripts/probe/priority_order_diag.py under
# runtime / library debug settings, one fresh process each.  usage: ... [outdir]
# RECORD of what was run in round 4 (profiles/r04_priority_env_sweep.txt): the CMDG_DBG_INIT /
# CMDG_DBG_EVRING / CMDG_DBG_EVFLAGS knobs it sets existed only in the library of commit 4a7157d and
# were removed with the fix; on the current library every line reports three OK trials.
OUT=${1:-gpurun_out/prio_env}
mkdir -p "$OUT"
: > "$OUT/summary.txt"
i=0
run() {
  i=$((i+1))
  log="$OUT/run_$i.log"
  env "$@" timeout -k 10 300 python scripts/probe/priority_order_diag.py $ARGS > "$log" 2>&1
  rc=$?
  echo "[$i] $* ARGS='$ARGS' rc=$rc | $(grep -c MISMATCH $log) of $(grep -c '^trial .: worst' $log) trials mismatch | $(grep 'worst' $log | tr '\n' ' ' | cut -c1-150)" | tee -a "$OUT/summary.txt"
}
ARGS=""
run CMDG_HALO_PRIORITY=1
run CMDG_HALO_PRIORITY=1 CMDG_DBG_INIT=1
run CMDG_HALO_PRIORITY=1 CMDG_DBG_INIT=2
run CMDG_HALO_PRIORITY=1 CMDG_DBG_INIT=4
run CMDG_HALO_PRIORITY=1 CMDG_DBG_INIT=8
run CMDG_HALO_PRIORITY=1 DIAG_WARM_PRIO_STREAMS=4
run CMDG_HALO_PRIORITY=1 DIAG_WARM_PRIO_STREAMS=4 DIAG_WARM_KEEP=1
run CMDG_HALO_PRIORITY=1 DIAG_WARM_PRIO_STREAMS=1
run CMDG_HALO_PRIORITY=1 AMD_LOG_LEVEL=0 HIP_LAUNCH_BLOCKING=0
run CMDG_HALO_PRIORITY=1
exit 0
