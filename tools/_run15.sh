cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/gpu_suite.log
b() { tag=$1; shift; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes , --no-side-runs --repeats 1 "$@" 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"])' $tag; }
b product-b3; b product-f16 --precision f16; BP_B3_PLANES=1 b product-b3-planes
b product-f16-b28 --precision f16 --batch 28 --streams 3 --steps 30
b product-b3-b28 --batch 28 --streams 2 --steps 30
