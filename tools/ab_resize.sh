#!/bin/bash
# Upper bound of fusing the two resize launches into the stem (round-4 verdict item 6): the headline pipeline with and without them, on the
# EXPERIMENTAL library (python -m betapose_amd.build --experimental; BP_ABLATE_RESIZE=1 leaves both launches out of the frame graph --
# a timing experiment with wrong results: the detector sees an unwritten input).   tools/ab_resize.sh [reps]
reps=${1:-3}
export BP_LIB=betapose_amd/libbetapose_hip_exp.so
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 3"
for rep in $(seq $reps); do
for on in 0 1; do
  if [ $on = 1 ]; then export BP_ABLATE_RESIZE=1; else unset BP_ABLATE_RESIZE; fi
  $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('no_resize=$on', d['value'], d.get('repeats', {}).get('fps'), 'nodes', d['config']['graph_nodes'])
"
done
done
