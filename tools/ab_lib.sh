#!/bin/bash
# A/B of two builds of the library on the configs[2] shape (28 frames per launch x 3 streams), one box: tools/ab_lib.sh <other .so> [reps] [precisions]
# alternates BP_LIB=<other .so> with the product library; frames/s, clock, socket power, joules per frame
LIB=$1; reps=${2:-2}; shift; shift
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes= --batch 28 --streams 3 --steps 60 --warmup 10 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('clocks_under_load', {})
        print('$1 | fps', d['value_settled'], d['value'], '| sclk', c.get('sclk_MHz_p50'), 'MHz | socket', c.get('package_W_p50'), 'W | J/frame', c.get('joules_per_frame'), '| limiters', c.get('limiters_active_fraction_of_samples'))
"; }
for rep in $(seq $reps); do for P in ${@:-f16r f16}; do
BP_LIB=$LIB $B --precision $P 2>/dev/null | val "$P $LIB"
$B --precision $P 2>/dev/null | val "$P product library"
done; done
