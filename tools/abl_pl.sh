#!/bin/bash
# Timing ablations of conv_pl.hip (experimental build): which part of the K loop bounds a layer.
#   tools/abl_pl.sh <mode b3|f16> <batch> <shape substring> <tiles>
cd "$(dirname "$0")/.."
export BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so
for abl in 0 1 2 3 4 12 7 15; do
  echo "== BP_PL_ABL=$abl (1 no activation DMA, 2 no filter DMA, 4 no MFMA, 8 no fragment reads)"
  BP_PL_ABL=$abl python tools/bench_pl.py --mode $1 --batch $2 --only $3 --tiles $4 --splits ${5:-1} 2>&1 | grep -v amdgpu.ids
done
