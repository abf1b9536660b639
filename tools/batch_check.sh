B="python bench.py --no-cpu-baseline --no-side-runs --no-served-legs --no-flip-rate --no-roofline --other-modes= --steps 60 --warmup 10 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], d['repeats']['fps'])
"; }
for pr in f16 f16r f16 f16r; do
$B --batch 28 --streams 3 --precision $pr 2>/dev/null | val ${pr}_b28s3
done
$B --steps 300 --precision f16 2>/dev/null | val f16_b1s4
$B --steps 300 --precision f16r 2>/dev/null | val f16r_b1s4
