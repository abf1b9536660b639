"""Pose solve (bp_solve_pnp, host C++ f64) -- known-answer tests (OpenCV is not available, so
cv2.solvePnP parity is pinned by synthetic poses and an independent scipy minimiser). CPU only."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from betapose_amd.ops import solve_pnp
from betapose_amd.synth import CAM_K, synth_kp3d
from oracle import post_ref


def _project(P, R, t):
    Y = P @ R.T + t
    uv = Y @ CAM_K.T
    return uv[:, :2] / uv[:, 2:]


def _poses(n, seed):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        R = Rot.from_rotvec(rng.normal(0, 0.9, 3)).as_matrix()
        t = np.array([rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1), rng.uniform(0.4, 1.5)])
        yield R, t, rng


@pytest.mark.parametrize("npts", [50, 10, 6])
def test_noise_free_pose_is_recovered(npts):
    P = synth_kp3d(50)[:npts]
    for R, t, _ in _poses(50, 1):
        R1, t1 = solve_pnp(P, _project(P, R, t), CAM_K)
        assert np.abs(R1 - R).max() < 1e-8 and np.abs(t1[:, 0] - t).max() < 1e-8
        assert abs(np.linalg.det(R1) - 1) < 1e-12


def test_noisy_pose_matches_independent_minimiser():
    P = synth_kp3d(50)
    for R, t, rng in _poses(40, 2):
        uv = _project(P, R, t) + rng.normal(0, 1.0, (50, 2))
        R1, t1 = solve_pnp(P, uv, CAM_K)
        R2, t2, _ = post_ref.pnp_least_squares(P, uv, CAM_K, R, t)   # scipy LM started at the truth
        assert np.abs(R1 - R2).max() < 1e-6 and np.abs(t1 - t2).max() < 1e-6


def test_designated_keypoints_from_reference_assets():
    """The only real data fixtures of the reference: 1_keypoint_designator/assets/sifts/*.ply (50 3-D key points in
    mm).  Build-container only -- the GPU box has no /root/reference."""
    ply = "/root/reference/1_keypoint_designator/assets/sifts/1.ply"
    if not os.path.exists(ply):
        pytest.skip("reference assets not present")
    lines = open(ply).read().split("\n")
    start = lines.index("end_header") + 1
    pts = np.array([[float(v) for v in ln.split()[:3]] for ln in lines[start:start + 50] if ln.strip()]) * 0.001
    assert pts.shape == (50, 3)
    for R, t, _ in _poses(10, 3):
        R1, t1 = solve_pnp(pts, _project(pts, R, t), CAM_K)
        assert np.abs(R1 - R).max() < 1e-7 and np.abs(t1[:, 0] - t).max() < 1e-7


def test_rejects_too_few_points():
    from betapose_amd._lib import BetaposeHipError
    P = synth_kp3d(5)
    with pytest.raises(BetaposeHipError):
        solve_pnp(P, np.zeros((5, 2)), CAM_K)
