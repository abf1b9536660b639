cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/bench_pl.py --batch 1 --mode b3 --tiles pl64 --engine-like --all-splits --splits 1,2,3,4,5,6,8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl5_engine_like.log
python tools/bench_pl.py --batch 1 --mode b3 --tiles pl64 --all-splits --splits 1,2,3,4,5,6,8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bpl5_plain.log
