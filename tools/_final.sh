#!/bin/bash
cd "$(dirname "$0")/.."
python bench.py > gpurun_out/r04_bench_default.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_protocol.json 2>/dev/null
bash tools/_energy_modes.sh > gpurun_out/r04_clocks_power.txt 2>&1
tail -c 400 gpurun_out/r04_bench_default.json; echo; cat gpurun_out/r04_clocks_power.txt | cut -c1-200
