#!/bin/bash
# configs[2] (28 frames per launch x 3 streams): clocks / socket power / limiter / joules per frame with and without the round-6 kernels, one box
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes= --batch 28 --streams 3 --steps 60 --warmup 10 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('clocks_under_load', {})
        print('$1 | fps', d['value_settled'], '| sclk', c.get('sclk_MHz_p50'), 'MHz | socket', c.get('package_W_p50'), 'W (energy counter:', c.get('mean_socket_W_from_energy'), 'W) | J/frame', c.get('joules_per_frame'), '| limiters', c.get('limiters_active_fraction_of_samples'), '| fps meanwhile', c.get('frames_per_sec_meanwhile'))
"; }
for P in f16r f16; do
BP_NO_P3=1 BP_NO_STEM7=1 $B --precision $P 2>/dev/null | val "$P, round-5 plan (BP_NO_P3=1 BP_NO_STEM7=1)"
BP_P3_K1=0 BP_NO_STEM7=1 $B --precision $P 2>/dev/null | val "$P, + conv_p3 on the 3x3 layers"
BP_NO_STEM7=1 $B --precision $P 2>/dev/null | val "$P, + its 1x1 form (K >= 512)"
$B --precision $P 2>/dev/null | val "$P, + fp16 7x7 stem (the plan)"
done
