B="python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 300 --warmup 60 --repeats 2"
val() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['value'], d['repeats']['fps'])
"; }
for thr in 64 32 16 8; do
export BP_HALO_BATCH_MIN_TILES=$thr
$B --batch 2 --streams 4 2>/dev/null | val b2s4_thr$thr
$B --batch 4 --streams 3 2>/dev/null | val b4s3_thr$thr
$B --batch 28 --streams 2 --steps 60 2>/dev/null | val b3b28_thr$thr
done
