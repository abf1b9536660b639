cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
tools/micro/run_dma_bw.sh 2>&1 | tee gpurun_out/dma_bw.txt
