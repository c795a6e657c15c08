#!/bin/bash
# A/B of one python command over environment settings; prints the last line of each run's stdout
# through a user-supplied python one-liner reading `d` (the parsed JSON).
# usage: scripts/ab_env_cmd.sh "<python cmd>" "<python expr on d>" "VAR=val" ... ("-" = none)
cmd="$1"; expr="$2"; shift 2
for kv in "$@"; do
  if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi
  env $pre python $cmd 2>/dev/null > /tmp/ab_env_cmd.out
  python -c "
import json
d = json.loads(open('/tmp/ab_env_cmd.out').read())
print('%-28s' % '$kv', $expr)
"
done
