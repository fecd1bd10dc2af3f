#!/bin/bash
# same-box A/B of two builds of the library on bench entries: tools/ab_c4.sh <libA> <libB> entry...
A=$1; B=$2; shift 2
for i in 1 2; do for lib in $A $B; do for s in "$@"; do
  BLR_MI355X_LIB=$PWD/$lib python bench.py --secondary-only $s 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())['secondary']
for k,v in d.items():
    print('$lib'[-16:], k, {x:v[x] for x in v if x in ('ms','ms_per_step','per_s','value','kernel_ms_median','frac')} or list(v)[:12])
"
done; done; done
