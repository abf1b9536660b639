cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "pl" 2>&1 | tail -5
python tools/bench_pl.py --batch 1 --mode b3 --tiles bd,pl64,pl128x64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl3_b3_1.log
python tools/bench_pl.py --batch 28 --mode f16 --splits 1 --only y3x3 --tiles pl64,pl128x64,pl128,pl256x128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl3_f16_28.log
python tools/bench_pl.py --batch 28 --mode b3 --splits 1 --only y3x3 --tiles pl64,pl128x64,pl128,pl256x128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl3_b3_28.log
timeout 1200 python -m pytest tests/test_gpu_nets.py -x -q 2>&1 | tail -15
