#!/bin/bash
# round-5 verdict item 6 as a per-layer plan: the small 3x3 / stride-1 layers of the bf16x3 mode on operand planes their producers write (BP_B3_MIX =
# largest map, pixels per image, that takes the plane path), A/B on one box against the default plan: frames/s, clock, joules per frame
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('clocks_under_load', {})
        print('$1 | fps', d['value_settled'], d['value'], '| sclk', c.get('sclk_MHz_p50'), 'MHz | socket', c.get('package_W_p50'), 'W | J/frame', c.get('joules_per_frame'), '| limiters', c.get('limiters_active_fraction_of_samples'))
"; }
for i in 1 2; do
$B 2>/dev/null | val "default plan"
BP_B3_MIX=320 $B 2>/dev/null | val "BP_B3_MIX=320 (20x16, 13x13, 10x8 3x3 layers on planes)"
BP_B3_MIX=100 $B 2>/dev/null | val "BP_B3_MIX=100 (10x8 only)"
BP_B3_MIX=1280 $B 2>/dev/null | val "BP_B3_MIX=1280 (+ 40x32, 26x26)"
done
