#!/bin/bash
# Timing experiment (wrong results, never shipped): rebuild the library with parts of the conv kernels compiled out and
# run the headline bench, to bound what removing the split-K hand-off / the K loop / the operand split / the MFMAs could
# buy.  Writes gpurun_out/ablate_pipeline.txt
mkdir -p gpurun_out
out=gpurun_out/ablate_pipeline.txt
: > $out
variants=("" "-DBP_ABLATE_TAIL" "-DBP_ABLATE_KLOOP" "-DBP_ABLATE_TAIL -DBP_ABLATE_KLOOP" "-DBP_ABLATE_SPLIT" "-DBP_ABLATE_MFMA" "-DBP_ABLATE_SPLIT -DBP_ABLATE_MFMA" "-DBP_ABLATE_PREFETCH" "-DBP_ABLATE_EPILOGUE" "-DBP_ABLATE_ACKWAIT" "-DBP_ABLATE_SLABSTORE" "-DBP_ABLATE_REDUCE")
for v in "${variants[@]}"; do
  # the ablations compile only into the experimental library (conv_dev.h refuses them otherwise); run the bench against it
  BP_CFLAGS="$v" python -m betapose_amd.build --force --experimental > /dev/null 2>&1 || { echo "build failed: $v" >> $out; continue; }
  export BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so
  for st in 4 1; do
    line=$(timeout 300 python bench.py --steps 400 --warmup 40 --streams $st --no-side-runs --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes "" --repeats 1 2>/dev/null | tail -1)
    echo "flags='$v' streams=$st $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("fps=%.1f ms_per_step=%.3f" % (d["value"], d["ms_per_step"]))')" >> $out
  done
done
unset BP_LIB
python -m betapose_amd.build --force --experimental > /dev/null 2>&1
cat $out
