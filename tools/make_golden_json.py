#!/usr/bin/env python3
"""Golden JSON texts from the REFERENCE's own ``write_json`` (3_6Dpose_estimator/pPose_nms.py:284-371) in its three
layouts -- the default list, 'cmu' and 'open' -- on seeded results (build container only; shims as tools/make_golden.py).
Writes tests/golden/json_formats.npz: the inputs as arrays and the files the reference wrote, as text."""
import json, os, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shims  # noqa: E402

ref_shims.install()
import torch  # noqa: E402

os.chdir(ref_shims.REF)
from opt import opt  # noqa: E402
import pPose_nms as ref_nms  # noqa: E402

g = np.random.Generator(np.random.PCG64(99))
names = ["0003.png", "0012.png", "seq_0042.png"]
per_image = [1, 2, 1]
kp = g.uniform(0, 640, (sum(per_image), 50, 2)).astype(np.float32)
sc = g.uniform(0, 1, (sum(per_image), 50, 1)).astype(np.float32)
prop = g.uniform(0.5, 3, sum(per_image)).astype(np.float32)
R = g.normal(size=(3, 3, 3)); t = g.normal(size=(3, 3, 1))
results, n = [], 0
for i, name in enumerate(names):
    humans = []
    for _ in range(per_image[i]):
        humans.append({"keypoints": torch.from_numpy(kp[n]), "kp_score": torch.from_numpy(sc[n]),
                       "proposal_score": torch.tensor([float(prop[n])])})
        n += 1
    results.append({"imgname": "some/dir/" + name, "result": humans, "cam_R": R[i] if i != 1 else [], "cam_t": t[i] if i != 1 else []})
out = {"names": np.array(names), "per_image": np.array(per_image), "kp": kp, "sc": sc, "prop": prop, "R": R, "t": t}
for form in (None, "cmu", "open"):
    for for_eval in ((False, True) if form is None else (False,)):   # (the reference itself fails on for_eval=True + 'cmu'/'open': int image ids have no .split)
        opt.format = form
        with tempfile.TemporaryDirectory() as d:
            ref_nms.write_json(results, d, for_eval=for_eval)
            key = "%s_%d" % (form or "default", int(for_eval))
            out["main_" + key] = np.array(open(os.path.join(d, "Betapose-results.json")).read())
            if form:
                sep = sorted(os.listdir(os.path.join(d, "sep-json")))
                out["sepnames_" + key] = np.array(sep)
                out["sep_" + key] = np.array([open(os.path.join(d, "sep-json", s)).read() for s in sep])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "json_formats.npz"), **out)
print("wrote tests/golden/json_formats.npz", [k for k in out])
