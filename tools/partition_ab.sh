#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 3"
run() { $B "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['value'], d.get('repeats', {}).get('fps'))
"; }
for rep in 1 2; do
run --streams 4
run --streams 4 --partition 4
run --streams 4 --partition 2
run --streams 8 --partition 4
run --streams 6 --partition 2
done
for pl in part_half part_third; do
export BP_PLAN_FILE=tools/plans/$pl.txt
echo "plan $pl:"
run --streams 4 --partition 4
run --streams 4
unset BP_PLAN_FILE
done
