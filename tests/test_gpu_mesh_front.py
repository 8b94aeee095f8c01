"""GPU parity of the mesher front end (src/sdf/meshers): ScalarSource / HermiteSource over unit-cube points
(isosurface.rs:78-99) and Mesh::postproc (mesh.rs:22-33) vs the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BOXES = [((-1, -1, -1), (1, 1, 1)), ((-1.0, -0.75, -1.25), (1.0, 1.0, 0.5)), ((0.1, 0.2, 0.3), (0.35, 0.9, 1.7))]


def unit_points(n, seed):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0.0, 1.0, size=(n, 3)).astype(np.float32)
    pts[:8] = [(0, 0, 0), (1, 1, 1), (0.5, 0.5, 0.5), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0.25, 0.75, 0.5), (1, 1, 0)]
    pts[8:40] = rng.integers(0, 65, size=(32, 3)).astype(np.float32) / np.float32(64)  # lattice corners of a 64-cell grid
    return pts


@pytest.mark.parametrize("sdf_id", [0, 1, 2])
@pytest.mark.parametrize("box", BOXES)
def test_scalar_and_hermite_sources_match_oracle(pkg, oracle, sdf_id, box):
    pts = unit_points(3000, 5)
    d = torch.from_numpy(pts).cuda()
    for kw in (dict(), dict(cube_half_side=0.4, sphere_radius=0.5, cube_material=1)):
        prm = pkg.default_params(**kw)
        oprm = oracle.params_from(prm)
        got = pkg.source_sample_scalar(prm, d, *box, sdf_id=sdf_id).cpu().numpy()
        np.testing.assert_array_equal(got.view(np.uint32), oracle.source_scalar_many(oprm, pts, *box, sdf_id).view(np.uint32))
        got_n = pkg.source_sample_normal(prm, d, *box, sdf_id=sdf_id).cpu().numpy()
        np.testing.assert_array_equal(got_n.view(np.uint32), oracle.source_normal_many(oprm, pts, *box, sdf_id).view(np.uint32))


def test_scalar_source_equals_world_space_sampling(pkg):
    """The unit-cube path is vert_pos_to followed by the ordinary distance-only sample."""
    pts = unit_points(1000, 9)
    lo, hi = np.float32(BOXES[1][0]), np.float32(BOXES[1][1])
    world = pts * (hi - lo) + lo
    prm = pkg.default_params()
    a = pkg.source_sample_scalar(prm, torch.from_numpy(pts).cuda(), *BOXES[1])
    b = pkg.sample_points(prm, torch.from_numpy(world).cuda(), True)[:, 0]
    assert torch.equal(a, b)


def make_vertices(n, seed):
    rng = np.random.default_rng(seed)
    v = np.zeros((n, 12), np.float32)
    v[:, 0:3] = rng.uniform(-1.1, 1.1, size=(n, 3)).astype(np.float32)
    # a third keep a mesher-provided normal, a third have none, the rest sit either side of |n|^2 = 1e-4
    v[: n // 3, 3:6] = rng.normal(size=(n // 3, 3)).astype(np.float32)
    k = n - 2 * (n // 3)
    v[-k:, 3] = np.float32(0.01) * (1 + rng.uniform(-1e-3, 1e-3, size=k)).astype(np.float32)
    v[:, 6:] = np.float32(-3.0)  # must be overwritten
    return v


@pytest.mark.parametrize("n", [1, 255, 256, 257, 5000])
@pytest.mark.parametrize("sdf_id", [0, 1, 2])
def test_mesh_postproc_matches_oracle(pkg, oracle, n, sdf_id):
    v = make_vertices(n, 100 + n)
    for kw in (dict(), dict(cube_material=1, sphere_material=0)):
        prm = pkg.default_params(**kw)
        got = pkg.mesh_postproc(prm, torch.from_numpy(v).cuda(), sdf_id=sdf_id).cpu().numpy()
        want = oracle.mesh_postproc(oracle.params_from(prm), v, sdf_id)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        np.testing.assert_array_equal(got[:, :3], v[:, :3])  # positions untouched


def test_mesh_postproc_unaligned_and_host_entry_point(pkg, oracle):
    prm = pkg.default_params()
    v = make_vertices(700, 3)
    want = oracle.mesh_postproc(oracle.params_from(prm), v)
    # a vertex array that starts 4 bytes into an allocation takes the scalar kernel
    raw = torch.zeros(700 * 12 + 1, dtype=torch.float32, device="cuda")
    view = raw[1:].view(700, 12)
    view.copy_(torch.from_numpy(v))
    pkg.check(pkg.lib.sdfv_mesh_postproc(C.byref(prm), 0, C.c_void_p(view.data_ptr()), 700, None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(view.cpu().numpy().view(np.uint32), want.view(np.uint32))
    host = v.copy()
    pkg.check(pkg.lib.sdfv_mesh_postproc_host(C.byref(prm), 0, host.ctypes.data, 700))
    np.testing.assert_array_equal(host.view(np.uint32), want.view(np.uint32))
    assert pkg.lib.sdfv_mesh_postproc(C.byref(prm), 9, C.c_void_p(view.data_ptr()), 700, None) == -2


def test_golden_mesh_front_fixture(pkg):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "mesh_front_200.npz"))
    keys = ["cube_half_side", "cube_material", "sphere_radius", "sphere_material", "max_distance_custom_material",
            "disable_sphere"]
    ints = {"cube_material", "sphere_material", "disable_sphere"}
    unit = torch.from_numpy(g["unit_points"]).cuda()
    box = (tuple(g["bb_min"]), tuple(g["bb_max"]))
    for k, row in enumerate(g["params"]):
        prm = pkg.default_params(**{kk: (int(v) if kk in ints else float(v)) for kk, v in zip(keys, row)})
        for sdf_id in (0, 1, 2):
            got = pkg.source_sample_scalar(prm, unit, *box, sdf_id=sdf_id).cpu().numpy()
            np.testing.assert_array_equal(got.view(np.uint32), g[f"scalar_{k}_{sdf_id}"].view(np.uint32))
            got = pkg.source_sample_normal(prm, unit, *box, sdf_id=sdf_id).cpu().numpy()
            np.testing.assert_array_equal(got.view(np.uint32), g[f"normal_{k}_{sdf_id}"].view(np.uint32))
            got = pkg.mesh_postproc(prm, torch.from_numpy(g["vertices"]).cuda(), sdf_id=sdf_id).cpu().numpy()
            np.testing.assert_array_equal(got.view(np.uint32), g[f"postproc_{k}_{sdf_id}"].view(np.uint32))


def test_empty_inputs_are_no_ops(pkg):
    """n = 0 everywhere in the mesher front end: success, nothing touched, no launch."""
    import ctypes as C
    prm = pkg.default_params()
    empty3 = torch.empty((0, 3), device="cuda")
    assert pkg.source_sample_scalar(prm, empty3).shape == (0,)
    assert pkg.source_sample_normal(prm, empty3).shape == (0, 3)
    assert pkg.mesh_postproc(prm, torch.empty((0, 12), device="cuda")).shape == (0, 12)
    assert pkg.lib.sdfv_mesh_postproc(C.byref(prm), 0, None, 0, None) == 0
    assert pkg.lib.sdfv_mesh_postproc_host(C.byref(prm), 0, None, 0) == 0
    lo, hi = pkg.f3((-1, -1, -1)), pkg.f3((1, 1, 1))
    assert pkg.lib.sdfv_source_sample_scalar(C.byref(prm), 0, lo, hi, None, 0, None, None) == 0
    assert pkg.lib.sdfv_source_sample_scalar(C.byref(prm), 0, None, hi, None, 0, None, None) == -1  # no bounding box
    assert pkg.lib.sdfv_source_sample_scalar(C.byref(prm), 0, lo, hi, None, 5, None, None) == -1    # n without buffers
