set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "pl" 2>&1 | tail -15 > gpurun_out/t_pl.log
cat gpurun_out/t_pl.log
timeout 300 python tools/bench_pl.py --batch 1 --mode b3 > gpurun_out/bpl_b3_1.log 2>&1; cat gpurun_out/bpl_b3_1.log
timeout 300 python tools/bench_pl.py --batch 28 --mode f16 --splits 1 --only y3x3 > gpurun_out/bpl_f16_28.log 2>&1; cat gpurun_out/bpl_f16_28.log
timeout 300 python tools/bench_pl.py --batch 28 --mode b3 --splits 1 --only y3x3 > gpurun_out/bpl_b3_28.log 2>&1; cat gpurun_out/bpl_b3_28.log
