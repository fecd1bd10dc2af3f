#!/bin/bash
# same-box A/B of a run-time switch on bench entries: tools/ab_env.sh VAR entry...   (runs each entry with VAR unset and VAR=1, twice)
V=$1; shift
for i in 1 2; do for on in 0 1; do for s in "$@"; do
  if [ $on = 1 ]; then export BLR_MI355X_$V=1; else unset BLR_MI355X_$V; fi
  python bench.py --secondary-only $s 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())['secondary']
for k,v in d.items():
    print('$V=$on', k, {x:v[x] for x in v if x in ('ms','per_s')}, 'frac', v.get('roofline',{}).get('frac'))
"
done; done; done
