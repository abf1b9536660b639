cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { tag=$1; shift; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes , --no-side-runs --repeats 1 "$@" 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])' $tag; }
e() { BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so BP_LEGACY=1 b "$@"; }
for r in 1 2; do b product-nst2; e round2-kernels; done
BP_SPLIT_PCT=75 b product-nst2-splits75
b product-nst2-1stream --streams 1
python tools/bench_pl.py --batch 1 --mode b3 --tiles pl64 --engine-like 2>&1 | grep -v amdgpu.ids
