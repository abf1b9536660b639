cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/tune_conv.py --pl 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tune_pl_b3.txt | tail -52
python tools/tune_conv.py --pl --f16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tune_pl_f16.txt | tail -3
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_stages.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','latency_ms','other_precisions')}, d['roofline']['isolated'])"
BP_KEEP_F32=1 timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes '' 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KEEP_F32', {k: d[k] for k in ('value','ms_per_step')})"
