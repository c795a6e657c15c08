mkdir -p gpurun_out/r4f
for st in torch system; do python scripts/probe/memset_null_stream_order.py $st > gpurun_out/r4f/memset_probe_$st.txt 2>&1; echo "memset probe $st rc=$?"; tail -2 gpurun_out/r4f/memset_probe_$st.txt; done
timeout -k 10 600 python -m pytest tests/test_gpu_split_explicit01.py tests/test_gpu_create_contract.py tests/test_gpu_split_explicit.py -x -q > gpurun_out/r4f/pytest_se.log 2>&1; echo pytest se rc=$?; tail -5 gpurun_out/r4f/pytest_se.log
# PMC for the paired tendency experiment (pairs build)
export CMDG_LIB=$PWD/build/libcmdg_pairs.so
CMDG_TENDENCY_PAIRS=1 PMC_FULLNAME=1 bash scripts/pmc.sh "--steps 3 --warmup 1 --no-parity --no-secondary" gpurun_out/r4f/pairs_hs_pmc.json "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/r4f/pairs_hs_pmc.log 2>&1; echo pmc hs rc=$?
CMDG_TENDENCY_PAIRS=1 PMC_FULLNAME=1 bash scripts/pmc.sh "--workload bomex --steps 3 --warmup 1 --no-parity --no-secondary" gpurun_out/r4f/pairs_bomex_pmc.json "FETCH_SIZE" "WRITE_SIZE" > gpurun_out/r4f/pairs_bomex_pmc.log 2>&1; echo pmc bomex rc=$?
unset CMDG_LIB
python - <<'PY'
import json
for n in ("hs", "bomex"):
    d = json.load(open("gpurun_out/r4f/pairs_%s_pmc.json" % n))
    for k, v in d.items():
        if isinstance(v, dict) and "k_tendency" in k:
            print(n, k[:90], {c: round(x, 1) for c, x in v.items()})
PY
# ocean timeline
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/octrace && (cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace -d /tmp/octrace -o t --output-format csv -- python3 bench.py --workload ocean-split-explicit --steps 3 --warmup 1 --no-cpu --no-events > /tmp/octrace.log 2>&1); f=$(find /tmp/octrace -name '*kernel_trace.csv' | head -1); cd $GRAFT_REPO_ROOT; python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernel records; columns:", list(rows[0].keys()))
import gzip, json
keep = [{k: r[k] for k in ("Kernel_Name", "Queue_Id", "Stream_Id", "Start_Timestamp", "End_Timestamp") if k in r} for r in rows]
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"].replace("void cmdg::", "").split("(")[0][:60]
json.dump(keep, gzip.open("gpurun_out/r4f/ocean_trace.json.gz", "wt"))
PY
