#!/bin/bash
# A/B of two builds of the library on tools/selfplay_bench.py, interleaved: tools/ab_selfplay.sh <other .so> [selfplay_bench flags]
other=$1; shift
for rep in 1 2; do
  for lib in "" "$other"; do
    GMK_HIP_LIB=$lib python3 tools/selfplay_bench.py "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('${lib:-production}'.split('/')[-1], round(d['games_per_s']), 'games/s', round(d['playouts_per_s']/1e6,1), 'M playouts/s', round(d['play_s'],2), 's')
"
  done
done
