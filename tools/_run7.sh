cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -5
python tools/bench_pl.py --batch 28 --mode f16 --splits 1 --only y3x3 --tiles pl64,pl128x64,pl128,pl256x128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl4_f16_28.log
python tools/bench_pl.py --batch 28 --mode b3 --splits 1 --only y3x3 --tiles pl128,pl256x128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl4_b3_28.log
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/gpu_suite.log
timeout 600 python bench.py --steps 100 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench_default.log
