#!/bin/bash
# A/B of plan files in the whole pipeline (4 frames in flight), alternating: tools/ab_plans.sh reps plan1 plan2 ...  ("none" = built-in plan)
reps=$1; shift
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 3"
for rep in $(seq $reps); do
for pl in "$@"; do
  if [ $pl = none ]; then unset BP_PLAN_FILE; else export BP_PLAN_FILE=tools/plans/$pl.txt; fi
  $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$pl', d['value'], d.get('repeats', {}).get('fps'))
"
done
done
