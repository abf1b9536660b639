#!/bin/bash
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --precision f16 "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %s -> %.1f fps' % (' '.join(sys.argv[1:]), d['value']))" "$@"; }
for S in 2 3 4 5 6; do run --steps 300 --warmup 30 --streams $S; done
run --steps 150 --warmup 15 --streams 4 --batch 2
run --steps 100 --warmup 10 --streams 3 --batch 4
run --steps 60 --warmup 6 --streams 2 --batch 8
run --steps 40 --warmup 4 --streams 2 --batch 28
run --steps 40 --warmup 4 --streams 3 --batch 28
run --steps 40 --warmup 4 --streams 2 --batch 28 --tile 1
