#!/bin/bash
amd-smi metric -g 0 --throttle --json 2>&1 | head -80
amd-smi metric -g 0 --energy --json 2>&1 | head -20
(python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --no-side-runs --no-roofline --other-modes= --steps 25000 --warmup 60 --repeats 1 > /dev/null 2>&1 &)
sleep 28
echo "== loaded"
amd-smi metric -g 0 --power --clock --json 2>&1 | head -60
amd-smi metric -g 0 --throttle 2>&1 | grep -v "N/A" | head -60
amd-smi metric -g 0 --energy --json 2>&1 | head -20
sleep 5
amd-smi metric -g 0 --throttle 2>&1 | grep "ACCUMULATED" -A1 | head -40
amd-smi metric -g 0 --energy --json 2>&1 | head -20
sleep 12
