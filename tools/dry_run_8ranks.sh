#!/bin/bash
# Round-5 verdict item 8: first contact with an 8-rank job BEFORE an 8-GPU node exists -- eight ranks under torch.distributed.run,
# collectives over gloo, all sharing the ONE GPU of the test box: rendezvous, weight broadcast, thread caps, record gather and the
# host cost per frame per rank (bench line: rccl.per_rank_host_cpu_ms_per_frame).  NOT a scaling number: eight processes share one
# device's four hardware queues.   tools/dry_run_8ranks.sh [ranks] > profiles/r05_dry_run_8ranks.txt
cd "$(dirname "$0")/.."
N=${1:-8}
export BP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
echo "# $N ranks x 1 GPU (gloo), bench.py --gpus $N --steps 40 --warmup 10"
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus $N --steps 40 --warmup 10 \
    --no-cpu-baseline --no-roofline --streams 2 2>/tmp/dry8.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['rccl']
        print('value', d['value'], 'frames/s aggregate (one shared GPU), ranks', r['ranks'], 'backend', r['backend'])
        print('weight_broadcast_ms', r['weight_broadcast_ms'], 'records_gathered', d['records_gathered'])
        print('per_rank_frames_per_sec', r['per_rank_frames_per_sec'])
        print('per_rank_host_cpu_ms_per_frame', r['per_rank_host_cpu_ms_per_frame'], 'host_cores', r['host_cores'])
"
tail -3 /tmp/dry8.err
echo "# evaluate.py --fused, $N ranks, 64 synthetic frames"
/usr/bin/time -v python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29712 evaluate.py --synthetic 64 \
    --outdir /tmp/bp_dry8 --fused 2>&1 | grep -E "ADD|frames|Elapsed|Maximum resident|error|Error" | head -12
