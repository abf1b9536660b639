"""CPU oracle for the Betapose per-frame inference hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain torch-CPU / numpy
fp32, what the reference computes on the path; each
function cites the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline -- never as the
product path.  ``betapose_amd`` must not import from here.

Parity pin: the restatement is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference's own Python
(``tools/make_golden.py``, run in the build container where /root/reference is
mounted; vectors under ``tests/golden/``).  Third-party arithmetic the
reference delegates to packages that are not in its tree is pinned as follows:
Pillow bicubic resize -> integer-exact against the installed Pillow;
torchsample ``SpecialCrop``/``Pad`` -> restated from its published behaviour
("parity unpinned" for that step); ``cv2.solvePnP`` -> OpenCV is
not installable here, PnP is pinned by known-answer tests only ("parity
unpinned" against OpenCV itself).
"""
