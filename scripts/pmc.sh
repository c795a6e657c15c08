#!/bin/bash
# PMC passes over the default bench (one --pmc set per pass, kernel trace only), then a
# per-kernel per-launch summary.  Usage: scripts/pmc.sh "<bench args>" out.json "SET1" "SET2" ...
# Run on the GPU box through gpurun; writes under gpurun_out/.
args="$1"; out="$2"; shift 2
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o p --output-format csv -- python3 $root/bench.py $args --no-cpu --no-events > /tmp/pmc_$i.log 2>&1 || { tail -5 /tmp/pmc_$i.log; exit 1; }
done
python3 - "$root/$out" "$i" <<'PY'
import csv, glob, json, os, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = collections.defaultdict(dict)
for i in range(1, n + 1):
    files = glob.glob('/tmp/pmc_%d/**/*counter_collection.csv' % i, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].replace('void cmdg::', '')
            # PMC_FULLNAME=1 keeps the template arguments (one entry per instantiation)
            k = k.split('(')[0][:110] if os.environ.get('PMC_FULLNAME') else k.split('<')[0].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k].add(r['Dispatch_Id'])
    for k in acc:
        for c, v in acc[k].items():
            res[k][c] = v / max(len(cnt[k]), 1)
        res[k]['launches'] = len(cnt[k])
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(out))))
    import bench
    res['kernel_source_digest'] = bench.kernel_source_digest()
except Exception as e:          # the counters stand without it; bench.py then calls them stale
    print('no kernel source digest:', e)
json.dump(res, open(out, 'w'), indent=1)
for k, d in res.items():
    if isinstance(d, dict):
        print(k, {c: round(v, 1) for c, v in d.items()})
PY
