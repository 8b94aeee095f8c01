"""GPU parity of batched SDFSurface::sample / ::normal (the per-point ABI's arithmetic) vs oracle + fixtures."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PARAM_KEYS = ["cube_half_side", "cube_material", "sphere_radius", "sphere_material",
              "max_distance_custom_material", "disable_sphere"]
INTS = {"cube_material", "sphere_material", "disable_sphere"}


def test_golden_points_fixture(pkg):
    g = np.load(os.path.join(GOLD, "points_512.npz"))
    pts = torch.from_numpy(g["points"]).cuda()
    for k, row in enumerate(g["params"]):
        prm = pkg.default_params(**{kk: (int(v) if kk in INTS else float(v)) for kk, v in zip(PARAM_KEYS, row)})
        for sdf_id in (0, 1, 2):
            for do in (0, 1):
                got = pkg.sample_points(prm, pts, bool(do), sdf_id).cpu().numpy()
                np.testing.assert_array_equal(got.view(np.uint32), g[f"s_{k}_{sdf_id}_{do}"].view(np.uint32))


def test_random_points_match_oracle(pkg, oracle):
    rng = np.random.default_rng(int(os.environ.get("SDFV_SOAK_SEED", 7)))  # tools/soak.sh varies the seed
    pts = rng.uniform(-1.5, 1.5, size=(20000, 3)).astype(np.float32)
    pts[:64] *= np.float32(1e-4)
    pts[64:128] *= np.float32(1e4)          # far away: fmod / floor on large arguments
    pts[128:160] = np.float32(0.0)
    pts[128:160, 0] = np.linspace(-1.2, 1.2, 32, dtype=np.float32)   # exactly on two axes' zero planes
    pts[160] = (0.0, 0.0, 0.0)              # NaN colour from the sphere's normalize, discarded or quantised to 0
    for kw in (dict(), dict(cube_material=1, sphere_material=0), dict(cube_half_side=0.3, sphere_radius=0.2)):
        prm = pkg.default_params(**kw)
        oprm = oracle.params_from(prm)
        for sdf_id in (0, 1, 2):
            got = pkg.sample_points(prm, torch.from_numpy(pts).cuda(), False, sdf_id).cpu().numpy()
            want = oracle.sample_many(oprm, pts, False, sdf_id)
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_normals_match_oracle(pkg, oracle):
    rng = np.random.default_rng(11)
    pts = rng.uniform(-1.2, 1.2, size=(4000, 3)).astype(np.float32)
    prm = pkg.default_params()
    oprm = oracle.params_from(prm)
    d = torch.from_numpy(pts).cuda()
    for sdf_id in (0, 1, 2):
        got = pkg.normal_points(prm, d, sdf_id=sdf_id).cpu().numpy()
        np.testing.assert_array_equal(got.view(np.uint32), oracle.normal_many(oprm, pts, sdf_id=sdf_id).view(np.uint32))
        for eps in (None, 0.01):  # normal_default_impl, defaults.rs:49-56
            got = pkg.normal_points(prm, d, eps=eps, use_default=True, sdf_id=sdf_id).cpu().numpy()
            want = oracle.normal_many(oprm, pts, eps=eps or 0.0, sdf_id=sdf_id, use_default=True)
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_empty_batch(pkg):
    prm = pkg.default_params()
    out = pkg.sample_points(prm, torch.empty((0, 3), device="cuda"))
    assert out.shape == (0, 7)


def test_unaligned_and_ragged_batches_take_the_scalar_path(pkg, oracle):
    """Pointers that are not 16-byte aligned and batch sizes that are not a multiple of the workgroup size."""
    rng = np.random.default_rng(23)
    prm = pkg.default_params()
    oprm = oracle.params_from(prm)
    for n in (1, 255, 256, 257, 1000):
        pts = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
        want = oracle.sample_many(oprm, pts)
        flat = torch.zeros(n * 3 + 1, device="cuda")
        flat[1:] = torch.from_numpy(pts).cuda().reshape(-1)
        for view in (flat[1:].view(n, 3), torch.from_numpy(pts).cuda()):   # misaligned by 4 bytes, aligned
            got = pkg.sample_points(prm, view).cpu().numpy()
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
