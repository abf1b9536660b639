B="python bench.py --no-cpu-baseline --no-side-runs --no-served-legs --no-flip-rate --no-roofline --other-modes= --steps 300 --warmup 60 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], d['repeats']['fps'])
"; }
$B --batch 2 --streams 4 2>/dev/null | val b2s4
BP_NO_HALO=1 $B --batch 2 --streams 4 2>/dev/null | val b2s4_nohalo
$B --batch 4 --streams 3 2>/dev/null | val b4s3
$B --batch 28 --streams 3 --precision f16 --steps 60 2>/dev/null | val f16b28
$B --batch 28 --streams 2 --steps 60 2>/dev/null | val b3b28
