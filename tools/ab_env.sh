#!/bin/bash
# A/B of an environment switch in the whole pipeline: tools/ab_env.sh reps VAR   (alternates VAR unset / VAR=1)
reps=$1; var=$2
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 3"
for rep in $(seq $reps); do
for on in 0 1; do
  if [ $on = 1 ]; then export $var=1; else unset $var; fi
  $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$var=$on', d['value'], d.get('repeats', {}).get('fps'), 'nodes', d['config']['graph_nodes'])
"
done
done
