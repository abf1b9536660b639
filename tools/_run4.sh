cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "pl" 2>&1 | tail -5
python tools/bench_pl.py --batch 1 --mode b3 --tiles bd,pl64,pl128x64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl2_b3_1.log
python tools/bench_pl.py --batch 28 --mode f16 --splits 1 --only y3x3 --tiles pl64,pl128x64,pl128,pl256x128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl2_f16_28.log
python tools/bench_pl.py --batch 28 --mode b3 --splits 1 --only y3x3 --tiles pl64,pl128x64,pl128,pl256x128 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl2_b3_28.log
BP_CONV_STAMPS=1 python tools/bench_pl.py --mode b3 --batch 1 --only y3x3_128 --tiles pl64 --splits 2 --iters 5 2>&1 | grep -v amdgpu.ids
BP_CONV_STAMPS=1 python tools/bench_pl.py --mode b3 --batch 28 --only y3x3_128 --tiles pl128 --splits 1 --iters 5 2>&1 | grep -v amdgpu.ids
BP_CONV_STAMPS=1 python tools/bench_pl.py --mode f16 --batch 28 --only y3x3_128 --tiles pl128 --splits 1 --iters 5 2>&1 | grep -v amdgpu.ids
