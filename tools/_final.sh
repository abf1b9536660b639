#!/bin/bash
# the round's last bench lines (default command, and the driver's protocol) -> gpurun_out/r04_bench_*.json
cd "$(dirname "$0")/.."
python bench.py > gpurun_out/r04_bench_default.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_protocol.json 2>/dev/null
python - <<'PY'
import json
for f in ['gpurun_out/r04_bench_default.json','gpurun_out/r04_bench_driver_protocol.json']:
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']; c=d['configs2']
    print(f, d['value'], d.get('value_settled'), 'frac', r['frac'], 'isolated', r['isolated']['frac'], 'configs2', c['value'], c['with_fp16_skip_connections']['value'])
PY
