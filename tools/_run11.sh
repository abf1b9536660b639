cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { tag=$1; shift; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes , --no-side-runs --repeats 1 "$@" 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])' $tag; }
e() { BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so BP_LEGACY=1 b "$@"; }
for r in 1 2 3; do b product; e round2-kernels; done
b product-1stream --streams 1; e round2-1stream --streams 1
b product-f16 --precision f16; e round2-f16 --precision f16
b product-f16-b28 --precision f16 --batch 28 --streams 3 --steps 30; e round2-f16-b28 --precision f16 --batch 28 --streams 3 --steps 30
b product-b3-b28 --batch 28 --streams 2 --steps 30; e round2-b3-b28 --batch 28 --streams 2 --steps 30
