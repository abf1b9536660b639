cd /tmp && export TMPDIR=/tmp
for L in y3x3_128_256_52 y3x3_32_64_s2_416; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcA_$L -o p -- python /root/repo/tools/bench_conv.py --iters 5 --only $L --splits 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmcC_$L -o p -- python /root/repo/tools/bench_conv.py --iters 5 --only $L --splits 1 > /dev/null 2>&1
done
