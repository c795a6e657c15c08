#!/bin/bash
# scripts/pmc_any.sh out.json "<python script and args>" "SET1" "SET2" ...: one rocprofv3 --pmc
# pass per counter set over the given python command, per-kernel per-launch averages.
out="$1"; cmd="$2"; shift 2
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmca_$i
  (cd $root && rocprofv3 --kernel-trace --pmc $set -d /tmp/pmca_$i -o p --output-format csv -- python3 $cmd > /tmp/pmca_$i.log 2>&1) || { tail -5 /tmp/pmca_$i.log; exit 1; }
done
python3 - "$root/$out" "$i" <<'PY'
import csv, glob, json, os, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = collections.defaultdict(dict)
for i in range(1, n + 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in glob.glob('/tmp/pmca_%d/**/*counter_collection.csv' % i, recursive=True):
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
    if k.startswith('k_') and isinstance(d, dict):
        print(k, {c: round(v, 1) for c, v in d.items()})
PY
