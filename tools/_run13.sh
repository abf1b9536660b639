cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/profile_ops.py --batch 28 2>&1 | grep -v amdgpu.ids > gpurun_out/ops_b28_product.txt
BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so BP_LEGACY=1 python tools/profile_ops.py --batch 28 2>&1 | grep -v amdgpu.ids > gpurun_out/ops_b28_round2.txt
python tools/profile_ops.py --batch 1 2>&1 | grep -v amdgpu.ids > gpurun_out/ops_b1_product.txt
BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so BP_LEGACY=1 python tools/profile_ops.py --batch 1 2>&1 | grep -v amdgpu.ids > gpurun_out/ops_b1_round2.txt
grep "^==" gpurun_out/ops_*.txt
