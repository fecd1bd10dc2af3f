#!/bin/bash
# same-box A/B of secondary bench entries with two libraries: tools/ab_sec.sh <entries> <libA|-> <libB|->
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
e=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset BLR_MI355X_LIB; else export BLR_MI355X_LIB=$R/$lib; fi
    echo "--- $lib (rep $rep)"
    python bench.py --secondary-only $e 2>/dev/null | grep '^{"secondary": "' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-28s %9.4f ms  frac %.3f  %s' % (d['secondary'], d['ms'], d['frac'], d['kernel'][:40]))"
  done
done
