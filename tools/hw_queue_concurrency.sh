#!/bin/bash
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 3000 --warmup 200 --repeats 1"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'])
"; }
echo "== 1 proc x 4 streams"; $B --streams 4 2>/dev/null | val p1s4
echo "== 2 procs x 4 streams"; ($B --streams 4 2>/dev/null | val p2s4a) & ($B --streams 4 2>/dev/null | val p2s4b) & wait
echo "== 2 procs x 2 streams"; ($B --streams 2 2>/dev/null | val p2s2a) & ($B --streams 2 2>/dev/null | val p2s2b) & wait
echo "== 2 procs x 3 streams"; ($B --streams 3 2>/dev/null | val p2s3a) & ($B --streams 3 2>/dev/null | val p2s3b) & wait
echo "== 1 proc x 8 streams, 8 hw queues"; GPU_MAX_HW_QUEUES=8 $B --streams 8 2>/dev/null | val q8s8
echo "== 1 proc x 6 streams, 6 hw queues"; GPU_MAX_HW_QUEUES=6 $B --streams 6 2>/dev/null | val q6s6
