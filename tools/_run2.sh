cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/abl_pl.sh f16 28 y3x3_128_256 pl128,pl128x64,pl256x128 > gpurun_out/abl_f16_28.log 2>&1
tools/abl_pl.sh b3 28 y3x3_128_256 pl64,pl128 > gpurun_out/abl_b3_28.log 2>&1
tools/abl_pl.sh b3 1 y3x3_128_256 pl64 2 > gpurun_out/abl_b3_1.log 2>&1
cat gpurun_out/abl_f16_28.log gpurun_out/abl_b3_28.log gpurun_out/abl_b3_1.log
