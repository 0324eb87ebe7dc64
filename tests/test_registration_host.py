"""Host logic of the registration estimate (biahub_amd/registration/ants.py) driven by the CPU oracle's metric.

The optimiser only sees an ``ops`` object (image_stats / smooth_shrink / mattes_mi); on the GPU box that is
``biahub_amd.registration.metric`` (HIP), here it is the oracle's NumPy restatement of the same three definitions,
so the multi-resolution ascent, the level geometry and the transform composition are tested without a GPU.
"""
import numpy as np
import pytest

from oracle import oracle_np as O


class OracleOps:
    @staticmethod
    def image_stats(v):
        st = O.image_stats(v)
        return {"min": st[0], "max": st[1], "sum": st[2], "center_of_mass": st[3:6] / st[2]}

    @staticmethod
    def smooth_shrink(v, sigma, factor):
        return O.smooth_shrink(v, sigma, factor)

    @staticmethod
    def mattes_mi(F, M, P, rng, bins=32, stride=1, offset=0):
        return O.mattes_mi(F, M, P, rng, bins, stride, offset)


def _similarity(angle_deg, scale, t):
    th = np.deg2rad(angle_deg)
    return np.array([[scale, 0, 0, t[0]], [0, scale * np.cos(th), -scale * np.sin(th), t[1]],
                     [0, scale * np.sin(th), scale * np.cos(th), t[2]], [0, 0, 0, 1.0]])


def test_mattes_gradient_matches_finite_differences():
    from scipy.ndimage import gaussian_filter

    A = gaussian_filter(O.synthetic_volume((24, 40, 48), seed=3, n_blobs=60), 1.0)
    B = O.affine_pull(A, _similarity(3.0, 1.01, (0.4, 1.5, -1.2)), A.shape, 1, O.BOUNDARY_ITK)
    Bc = B[4:20, 6:34, 6:42]  # every sample stays inside A for all nearby P: no validity jumps
    P = np.hstack([np.eye(3), np.array([[4.0], [6.0], [6.0]])])
    rng = (B.min(), B.max(), A.min(), A.max())
    v, g, n = O.mattes_mi(Bc, A, P, rng)
    assert n == Bc.size and v > 0.5
    fd = np.zeros((3, 4))
    for i in range(3):
        for j in range(4):
            Pp, Pm = P.copy(), P.copy()
            Pp[i, j] += 1e-4
            Pm[i, j] -= 1e-4
            fd[i, j] = (O.mattes_mi(Bc, A, Pp, rng)[0] - O.mattes_mi(Bc, A, Pm, rng)[0]) / 2e-4
    assert np.abs(fd - g).max() <= 0.08 * np.abs(g).max()


def test_smooth_shrink_geometry_and_constant():
    v = np.full((13, 20, 31), 7.0, np.float32)
    out, off = O.smooth_shrink(v, (2, 1, 0), (3, 2, 1))
    assert out.shape == (4, 10, 31) and off == (1, 0, 0)
    assert np.allclose(out, 7.0, rtol=1e-6)  # normalised kernel, edge clamp


@pytest.mark.parametrize("kind", ["Similarity", "Rigid"])
def test_optimiser_recovers_known_transform(kind):
    from biahub_amd.registration.ants import _optimise

    shape = (32, 96, 96)
    vol = O.synthetic_volume(shape, seed=5, n_blobs=150)
    M = _similarity(2.0, 1.02 if kind == "Similarity" else 1.0, (0.7, -2.25, 3.75))
    B = O.affine_pull(vol, M, shape, 1, O.BOUNDARY_ITK)  # B(p) = vol(M p): registering vol onto B must give M
    T = _optimise(B, vol, OracleOps, kind, (4, 2, 1), (400, 200, 30), (2, 1, 0), 32, 5, 0.25, False)
    assert np.abs(T[:3, :3] - M[:3, :3]).max() < 2e-3
    centre = (np.array(shape) - 1) / 2
    assert np.linalg.norm(T[:3] @ np.append(centre, 1) - M[:3] @ np.append(centre, 1)) < 0.1  # voxels, at the centre


def test_optimiser_planar_translation():
    from biahub_amd.registration.ants import _optimise

    img = O.synthetic_volume((1, 128, 128), seed=8, n_blobs=60)
    M = _similarity(0.0, 1.0, (0.0, 2.5, -1.75))
    B = O.affine_pull(img, M, img.shape, 1, O.BOUNDARY_ITK)
    T = _optimise(B, img, OracleOps, "Translation", (2, 1), (200, 50), (1, 0), 32, 5, 0.25, True)
    assert np.allclose(T[:3, :3], np.eye(3)) and T[0, 3] == 0.0
    assert np.abs(T[1:3, 3] - M[1:3, 3]).max() < 0.1


def test_postprocess_transform_composition():
    from biahub_amd.core.transform import Transform
    from biahub_amd.registration.ants import postprocess_transform

    init = Transform(_similarity(5.0, 1.1, (1, 2, 3)))
    fwd = Transform(_similarity(-1.0, 0.98, (0.5, 0.25, -0.75)))
    off = np.array([3.0, 10.0, 20.0])
    got = postprocess_transform(init, fwd, off).matrix
    to_roi, back = np.eye(4), np.eye(4)
    to_roi[:3, 3], back[:3, 3] = off, -off
    assert np.allclose(got, init.matrix @ to_roi @ fwd.matrix @ back)
    # a point of the cropped reference grid maps through fwd in ROI coordinates, then through the initial guess
    q = np.array([4.0, 5.0, 6.0, 1.0])
    assert np.allclose(got @ (q + np.append(off, 0)), init.matrix @ (np.append(off, 0) + fwd.matrix @ q))
