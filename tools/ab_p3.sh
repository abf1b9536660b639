#!/bin/bash
# A/B of the persistent 3x3 kernel (conv_p3.hip, TILE_P3) on the configs[2] shape (28 frames per launch x 3 streams): tools/ab_p3.sh reps [precisions]
# alternates BP_NO_P3=1 (the halo plane tile as before) with the default plan on ONE box; prints frames/s of every region
reps=${1:-2}; shift
B="python bench.py --batch 28 --streams 3 --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 60 --warmup 10 --repeats 3"
for rep in $(seq $reps); do
for prec in ${@:-f16r f16}; do
for off in 1 0; do
  if [ $off = 1 ]; then export BP_NO_P3=1; else unset BP_NO_P3; fi
  $B --precision $prec 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$prec', 'plh  ' if '$off' == '1' else 'p3   ', d['value'], d.get('repeats', {}).get('fps'), 'poses', d.get('poses'))
"
done
done
done
