#!/usr/bin/env python3
"""Occlusion-LineMod harness -- counterpart of the reference's ``occlusion_betapose_evaluate.py`` (run as in
``occlusion.sh``: ``--nClasses 50 --indir ... --outdir ... --sp --profile --conf 0.9 --obj_id N``).  Same pipeline as
evaluate.py with the occlusion protocol switched on: ground truth of sequence 02 (several objects per frame), the
``--left_keypoints`` highest-scoring key points for PnP, 20 px reprojection threshold."""
import sys

import evaluate

if __name__ == "__main__":
    if "--occlusion" not in sys.argv:
        sys.argv.append("--occlusion")
    evaluate.main()
