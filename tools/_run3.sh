cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in pl128 pl128x64 pl256x128 pl64; do
BP_CONV_STAMPS=1 python tools/bench_pl.py --mode f16 --batch 28 --only y3x3_128_256 --tiles $t --splits 1 --iters 5 2>&1 | grep -v amdgpu.ids
done > gpurun_out/stamps_f16_28.log
for t in pl128 pl64; do
BP_CONV_STAMPS=1 python tools/bench_pl.py --mode b3 --batch 28 --only y3x3_128_256 --tiles $t --splits 1 --iters 5 2>&1 | grep -v amdgpu.ids
done > gpurun_out/stamps_b3_28.log
for t in pl64; do
BP_CONV_STAMPS=1 python tools/bench_pl.py --mode b3 --batch 1 --tiles $t --splits 2,5 --iters 5 2>&1 | grep -v amdgpu.ids
done > gpurun_out/stamps_b3_1.log
cat gpurun_out/stamps_f16_28.log gpurun_out/stamps_b3_28.log gpurun_out/stamps_b3_1.log
