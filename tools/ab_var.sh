#!/bin/bash
# A/B of an environment switch on the configs[2] shape (28 frames per launch x 3 streams): tools/ab_var.sh reps precision VAR  (alternates VAR=1 / unset)
reps=${1:-2}; prec=${2:-f16r}; var=$3
B="python bench.py --batch 28 --streams 3 --precision $prec --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 60 --warmup 10 --repeats 3"
for rep in $(seq $reps); do
for on in 1 0; do
  if [ $on = 1 ]; then export $var=1; else unset $var; fi
  $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$prec $var=$on', d['value'], d.get('repeats', {}).get('fps'), 'poses', d.get('poses'))
"
done
done
