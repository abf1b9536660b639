#!/bin/bash
# Timing-ablation variants of the persistent 3x3 kernel (conv_p3.hip, -DP3_ABL=n: parts compiled out, WRONG results): conv_p3.o rebuilt per
# variant and linked with the product library's other objects into betapose_amd/libbetapose_hip_abl<n>.so (git-ignored; travels with gpurun).
# usage: tools/p3_variants.sh 1 2 4 ...     then on the GPU box: BP_LIB=betapose_amd/libbetapose_hip_abl4.so python tools/bench_p3.py
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd $REPO/betapose_amd
OBJS=$(ls build/*.o | grep -v conv_p3)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c -I csrc csrc/conv_p3.hip -DP3_ABL=$n $P3_EXTRA -o build_exp/conv_p3_abl$n.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libbetapose_hip_abl$n.so $OBJS build_exp/conv_p3_abl$n.o -lz -lpthread && echo built libbetapose_hip_abl$n.so
done
