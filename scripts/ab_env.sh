#!/bin/bash
# A/B over environment settings: scripts/ab_env.sh "<bench args>" "VAR=val" "VAR=val2" ... ("-" = none)
args="$1"; shift
for kv in "$@"; do
  if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi
  env $pre python bench.py $args --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels_ms']
print('%-28s' % '$kv', 'ms/step %.3f' % d['ms_per_step'], 'value %.3e' % d['value'], ' '.join('%s=%.1fus' % (n[:5], v['avg_ms'] * 1e3) for n, v in k.items()), 'roof %.3f' % d['roofline']['frac'])
"
done
