#!/bin/bash
# the fp16 7x7 stem in the configs[2] pipeline, A/B on one box (BP_NO_STEM7=1: the fp32 MFMA kernel as before): frames/s, clock, joules per frame
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes= --batch 28 --streams 3 --steps 60 --warmup 10 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('clocks_under_load', {})
        print('$1 | fps', d['value_settled'], d['value'], '| sclk', c.get('sclk_MHz_p50'), 'MHz | socket', c.get('package_W_p50'), 'W | J/frame', c.get('joules_per_frame'), '| limiters', c.get('limiters_active_fraction_of_samples'))
"; }
for i in 1 2; do for P in f16r f16; do
BP_NO_STEM7=1 $B --precision $P 2>/dev/null | val "$P BP_NO_STEM7=1"
$B --precision $P 2>/dev/null | val "$P fp16 stem"
done; done
