"""The device mesher (sdfv_mesh_extract): the extraction algorithm is the build's own (the reference delegates to the
un-vendored `isosurface` crate), so the checks are (1) an independent numpy restatement of the same conventions
(tools/gen_mc_table.py documents them) fed with the ORACLE's lattice distances -- vertices and indices must match
exactly -- and (2) properties any correct extractor has: closed 2-manifold, outward orientation, Euler
characteristic, vertices on the surface."""
import importlib.util
import os
from collections import Counter

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


@pytest.fixture(scope="module")
def table():
    spec = importlib.util.spec_from_file_location("gen_mc_table", os.path.join(ROOT, "tools", "gen_mc_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def lattice_distances(oracle, prm, n, box, sdf_id):
    ax = (np.arange(n + 1, dtype=np.float32) / F(n)).astype(np.float32)
    pts = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1)  # [i, j, k, 3]
    d = oracle.source_scalar_many(prm, pts.reshape(-1, 3), box[0], box[1], sdf_id).reshape(n + 1, n + 1, n + 1)
    return ax, d  # d[i, j, k]


def numpy_extract(table, oracle, prm, n, box, sdf_id):
    """Restatement of the extractor: lattice order (k, j, i) then axis order for vertices, cell order for triangles."""
    ax, d = lattice_distances(oracle, prm, n, box, sdf_id)
    inside = d < 0
    lo, size = np.float32(box[0]), np.float32(box[1]) - np.float32(box[0])
    vid, verts = {}, []
    for k in range(n + 1):
        for j in range(n + 1):
            for i in range(n + 1):
                idx = (i, j, k)
                for a in range(3):
                    nb = list(idx)
                    nb[a] += 1
                    if nb[a] > n or inside[tuple(nb)] == inside[idx]:
                        continue
                    d0, d1 = d[idx], d[tuple(nb)]
                    t = d0 / (d0 - d1)
                    u = [ax[idx[0]], ax[idx[1]], ax[idx[2]]]
                    u[a] = u[a] + t * (ax[idx[a] + 1] - u[a])
                    vid[(idx, a)] = len(verts)
                    verts.append([u[b] * size[b] + lo[b] for b in range(3)])
    tris = []
    for k in range(n):
        for j in range(n):
            for i in range(n):
                case = 0
                for c in range(8):
                    if inside[i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1)]:
                        case |= 1 << c
                for tri in table[case]:
                    for e in tri:
                        a, s = divmod(e, 4)
                        others = [b for b in range(3) if b != a]
                        owner = [i, j, k]
                        owner[others[0]] += s & 1
                        owner[others[1]] += s >> 1
                        tris.append(vid[(tuple(owner), a)])
    return np.array(verts, np.float32).reshape(-1, 3), np.array(tris, np.int64), d


@pytest.mark.parametrize("sdf_id,kw,box,n", [
    (2, dict(sphere_radius=0.8), ((-1, -1, -1), (1, 1, 1)), 12),
    (0, dict(), ((-1, -1, -1), (1, 1, 1)), 16),
    (1, dict(cube_half_side=0.6), ((-1.0, -0.75, -1.25), (1.0, 1.0, 0.5)), 9),
    (0, dict(cube_half_side=0.7, sphere_radius=0.8), ((-1, -1, -1), (1, 1, 1)), 11),
])
def test_extraction_matches_numpy_restatement(pkg, oracle, table, sdf_id, kw, box, n):
    prm = pkg.default_params(**kw)
    oprm = oracle.params_from(prm)
    v, idx = pkg.mesh_extract(prm, n, *box, sdf_id=sdf_id)
    want_v, want_i, _ = numpy_extract(table, oracle, oprm, n, box, sdf_id)
    assert v.shape[0] == want_v.shape[0] and idx.shape[0] == want_i.shape[0]
    got = v.cpu().numpy()
    np.testing.assert_array_equal(got[:, :3].view(np.uint32), want_v.view(np.uint32))
    np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int64), want_i)
    # normals = HermiteSource at the vertex, material fields = Vertex::default()
    want_n = oracle.normal_many(oprm, want_v, 0.0, sdf_id)
    np.testing.assert_array_equal(got[:, 3:6].view(np.uint32), want_n.view(np.uint32))
    assert (got[:, 6:] == 0).all()


def manifold_report(indices):
    tri = indices.reshape(-1, 3)
    directed = Counter()
    for a, b, c in tri:
        for e in ((a, b), (b, c), (c, a)):
            directed[e] += 1
    assert all(cnt == 1 for cnt in directed.values()), "an oriented edge is used twice"
    assert all((b, a) in directed for (a, b) in directed), "an edge has no opposite partner: the mesh is open"
    return len(directed) // 2


@pytest.mark.parametrize("n", [16, 33, 64])
def test_sphere_is_a_closed_oriented_genus_0_surface(pkg, oracle, n):
    prm = pkg.default_params(sphere_radius=0.8)
    v, idx = pkg.mesh_extract(prm, n, sdf_id=2)
    v, idx = v.cpu().numpy(), idx.cpu().numpy().astype(np.int64)
    n_edges = manifold_report(idx)
    assert v.shape[0] - n_edges + idx.shape[0] // 3 == 2  # Euler characteristic of a sphere
    assert idx.min() == 0 and idx.max() == v.shape[0] - 1 and len(np.unique(idx)) == v.shape[0]
    r = np.linalg.norm(v[:, :3].astype(np.float64), axis=1)
    assert np.abs(r - 0.8).max() < (2.0 / n) ** 2  # linear interpolation of an exact distance: O(h^2) off the sphere
    tri = v[idx.reshape(-1, 3), :3].astype(np.float64)
    face_n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    centroid = tri.mean(axis=1)
    assert (np.einsum("ij,ij->i", face_n, centroid) > 0).all()  # counter-clockwise seen from outside
    assert (np.einsum("ij,ij->i", v[:, 3:6], v[:, :3]) > 0).all()  # Hermite normals point outward


def test_demo_sdf_mesh_is_closed_and_postproc_fills_materials(pkg, oracle):
    prm = pkg.default_params()
    v, idx = pkg.mesh_extract(prm, 48)
    manifold_report(idx.cpu().numpy().astype(np.int64))
    before = v.clone()
    pkg.mesh_postproc(prm, v)
    want = oracle.mesh_postproc(oracle.params_from(prm), before.cpu().numpy())
    np.testing.assert_array_equal(v.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert (v[:, 6:9].cpu().numpy() > 0).any()


def test_empty_surface_bad_algorithm_and_limits(pkg):
    import ctypes as C
    prm = pkg.default_params(sphere_radius=5.0)
    v, idx = pkg.mesh_extract(prm, 8, sdf_id=2)  # everything inside: no crossing
    assert v.shape == (0, 12) and idx.shape == (0,)
    m = pkg._capi.Mesh()
    lo, hi = pkg.f3((-1, -1, -1)), pkg.f3((1, 1, 1))
    assert pkg.lib.sdfv_mesh_extract(C.byref(prm), 0, lo, hi, 8, 3, C.byref(m), None) == -1
    assert b"Unsupported algorithm" in pkg.lib.sdfv_last_error()
    assert pkg.lib.sdfv_mesh_extract(C.byref(prm), 0, lo, hi, 0, 0, C.byref(m), None) == -1
    assert pkg.lib.sdfv_mesh_extract(C.byref(prm), 0, lo, hi, 4096, 0, C.byref(m), None) == -1
    assert pkg.lib.sdfv_mesh_free(C.byref(m)) == 0 and pkg.lib.sdfv_mesh_free(None) == 0


def test_large_lattice_counts(pkg):
    """512^3 cells: 135 M lattice points through both scans; the sphere's triangle count scales with area."""
    prm = pkg.default_params(sphere_radius=0.8)
    v, idx = pkg.mesh_extract(prm, 512, sdf_id=2)
    tri = idx.shape[0] // 3
    assert v.shape[0] - tri * 3 // 2 + tri == 2  # closed triangle mesh: E = 3F/2
    area_cells = 4 * np.pi * (0.8 * 256) ** 2
    assert 1.5 * area_cells < tri < 3.0 * area_cells


def test_scratch_is_reused_and_can_be_trimmed(pkg):
    prm = pkg.default_params()
    a = pkg.mesh_extract(prm, 40)
    free_before = torch.cuda.mem_get_info()[0]
    b = pkg.mesh_extract(prm, 24)          # smaller: runs inside the kept scratch
    assert torch.cuda.mem_get_info()[0] >= free_before - (8 << 20)
    c = pkg.mesh_extract(prm, 40)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and b[0].shape[0] < a[0].shape[0]
    assert pkg.lib.sdfv_mesh_trim() == 0 and pkg.lib.sdfv_mesh_trim() == 0
    d = pkg.mesh_extract(prm, 40)
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])


def test_randomised_extractions_match_numpy_restatement(pkg, oracle, table):
    """Seeded sweep (tools/soak.sh varies the seed): random demo parameters, sub-trees, boxes and lattice sizes."""
    rng = np.random.default_rng(int(os.environ.get("SDFV_SOAK_SEED", 11)))
    for _ in range(int(os.environ.get("SDFV_SOAK_TRIALS", 4))):
        kw = dict(cube_half_side=float(np.float32(rng.uniform(0.2, 1.0))), sphere_radius=float(np.float32(rng.uniform(0.2, 1.2))),
                  disable_sphere=int(rng.integers(0, 4) == 0))
        lo = rng.uniform(-1.4, -0.6, size=3).astype(np.float32)
        hi = (lo + rng.uniform(1.2, 2.8, size=3)).astype(np.float32)
        box = (tuple(float(x) for x in lo), tuple(float(x) for x in hi))
        n, sdf_id = int(rng.integers(3, 14)), int(rng.integers(0, 3))
        prm = pkg.default_params(**kw)
        v, idx = pkg.mesh_extract(prm, n, *box, sdf_id=sdf_id)
        want_v, want_i, _ = numpy_extract(table, oracle, oracle.params_from(prm), n, box, sdf_id)
        assert v.shape[0] == want_v.shape[0], (kw, box, n, sdf_id)
        if v.shape[0]:
            np.testing.assert_array_equal(v.cpu().numpy()[:, :3].view(np.uint32), want_v.view(np.uint32))
            np.testing.assert_array_equal(idx.cpu().numpy().astype(np.int64), want_i)
