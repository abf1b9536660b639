#!/bin/bash
# clocks / power / throttle reason / joules per frame per mode (bench.py clocks_under_load), one box
B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes= --steps 200 --warmup 60 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d.get('clocks_under_load', {})
        print('$1 | fps', d['value_settled'], '| sclk', c.get('sclk_MHz_p50'), 'MHz | socket', c.get('package_W_p50'), 'W (energy counter:', c.get('mean_socket_W_from_energy'), 'W) | J/frame', c.get('joules_per_frame'), '| limiters', c.get('limiters_active_fraction_of_samples'), '| fps meanwhile', c.get('frames_per_sec_meanwhile'))
"; }
$B 2>/dev/null | val "bf16x3 (default: halo + filters-direct), 4 in flight"
BP_NO_HALO=1 $B 2>/dev/null | val "bf16x3 round-3 plan (BP_NO_HALO=1), 4 in flight"
BP_B3_PLANES=1 $B 2>/dev/null | val "bf16x3 on operand planes, 4 in flight"
$B --precision f32 2>/dev/null | val "fp32 MFMA, 4 in flight"
$B --precision f16 2>/dev/null | val "fp16, 4 in flight"
$B --streams 1 2>/dev/null | val "bf16x3, one frame at a time"
$B --streams 2 2>/dev/null | val "bf16x3, 2 in flight"
$B --batch 28 --streams 2 --steps 60 2>/dev/null | val "bf16x3, batch 28 x 2 streams"
$B --batch 28 --streams 3 --steps 60 --precision f16 2>/dev/null | val "fp16, batch 28 x 3 streams"
$B --batch 28 --streams 3 --steps 60 --precision f16r 2>/dev/null | val "fp16 with fp16 skip connections (f16r), batch 28 x 3 streams"
