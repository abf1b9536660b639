#!/bin/bash
# A/B of plan files on the configs[2] shape (28 frames per launch x 3 streams): tools/ab28.sh reps precision plan1 plan2 ... ("none" = built-in)
reps=$1; prec=$2; shift; shift
B="python bench.py --batch 28 --streams 3 --precision $prec --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 60 --warmup 10 --repeats 3"
for rep in $(seq $reps); do
for pl in "$@"; do
  if [ $pl = none ]; then unset BP_PLAN_FILE; else export BP_PLAN_FILE=tools/plans/$pl.txt; fi
  $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$prec $pl', d['value'], d.get('repeats', {}).get('fps'), 'poses', d.get('poses'))
"
done
done
