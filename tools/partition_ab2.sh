#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 3"
run() { $B "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('${BP_PLAN_FILE:-builtin} $*', d['value'], d.get('repeats', {}).get('fps'))
"; }
for rep in 1 2; do
unset BP_PLAN_FILE; run --streams 4; run --streams 4 --partition 2
export BP_PLAN_FILE=tools/plans/part_34.txt; run --streams 4 --partition 2
export BP_PLAN_FILE=tools/plans/part_half.txt; run --streams 4 --partition 2
done
