#!/bin/bash
# frames/s with each in-flight frame on its own CU slice: partition count x split-K target sweep
cd "$(dirname "$0")/.."
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  value %.1f fps  ms/step %.4f' % (d['value'], d['ms_per_step']))"; }
echo "baseline 4 streams shared"; run --streams 4
for P in 2 4 8; do
  for T in 512 256 128 64; do
    echo "partition $P streams $P sk-target $T"; run --streams $P --partition $P --sk-target $T
  done
done
echo "partition 4, 8 streams (2 per slice), target 128"; run --streams 8 --partition 4 --sk-target 128
echo "partition 8, 16 streams"; run --streams 16 --partition 8 --sk-target 64
