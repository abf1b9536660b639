#!/bin/bash
# A/B of an environment setting on the configs[2] shape (28 frames per launch x 3 streams), alternating on ONE box:
#   tools/ab28_env.sh reps "VAR=a" "VAR=b" [precisions]      e.g. tools/ab28_env.sh 2 BP_P3_K1=0 BP_P3_K1=1 f16r f16
reps=$1; A=$2; Bv=$3; shift 3
CMD="python bench.py --batch 28 --streams 3 --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 60 --warmup 10 --repeats 3"
for rep in $(seq $reps); do
for prec in ${@:-f16r}; do
for setting in "$A" "$Bv"; do
  env $setting $CMD --precision $prec 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$prec', '$setting', d['value'], d.get('repeats', {}).get('fps'), 'poses', d.get('poses'))
"
done
done
done
