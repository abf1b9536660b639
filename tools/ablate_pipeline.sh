#!/bin/bash
# Timing experiment (wrong results, never shipped): rebuild the library with parts of the conv kernels compiled out and
# run the headline bench, to bound what removing the split-K hand-off / the K loop / the operand split / the MFMAs could
# buy.  Writes gpurun_out/ablate_pipeline.txt
mkdir -p gpurun_out
out=gpurun_out/ablate_pipeline.txt
: > $out
variants=("" "-DBP_ABLATE_TAIL" "-DBP_ABLATE_KLOOP" "-DBP_ABLATE_TAIL -DBP_ABLATE_KLOOP" "-DBP_ABLATE_SPLIT" "-DBP_ABLATE_MFMA" "-DBP_ABLATE_SPLIT -DBP_ABLATE_MFMA" "-DBP_ABLATE_PREFETCH" "-DBP_ABLATE_EPILOGUE" "-DBP_ABLATE_ACKWAIT" "-DBP_ABLATE_SLABSTORE" "-DBP_ABLATE_REDUCE")
for v in "${variants[@]}"; do
  BP_CFLAGS="$v" python -m betapose_amd.build --force > /dev/null 2>&1 || { echo "build failed: $v" >> $out; continue; }
  for st in 4 1; do
    line=$(timeout 300 python bench.py --steps 400 --warmup 40 --streams $st --no-side-runs --no-cpu-baseline --no-roofline --other-modes "" 2>/dev/null | tail -1)
    echo "flags='$v' streams=$st $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("fps=%.1f ms_per_step=%.3f" % (d["value"], d["ms_per_step"]))')" >> $out
  done
done
python -m betapose_amd.build --force > /dev/null 2>&1
cat $out
