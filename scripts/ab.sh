#!/bin/bash
# A/B harness for tuning builds: scripts/ab.sh "<bench args>" lib1 lib2 ...
args="$1"; shift
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset CMDG_LIB; else export CMDG_LIB=$PWD/build/libcmdg_$lib.so; fi
  python bench.py $args --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels_ms']
print('%-10s' % '$lib', 'ms/step %.3f' % d['ms_per_step'], 'value %.3e' % d['value'], ' '.join('%s=%.1fus' % (n[:4], v['avg_ms'] * 1e3) for n, v in k.items()), 'roof %.3f' % d['roofline']['frac'])
"
done
