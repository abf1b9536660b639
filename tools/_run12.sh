cd $GRAFT_REPO_ROOT
r() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes , --no-side-runs --repeats 1 "$@" 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["roofline"]["isolated"]); [print("   ", k, v) for k, v in d["roofline"]["layer_classes"].items()]' X; }
echo PRODUCT; r
echo ROUND2; BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so BP_LEGACY=1 r
