"""BASELINE.json configs[1] and configs[2] at full size, every voxel and every pixel against the oracle (the C oracle
fills 512^3 in about a second on the box's cores): textures and the pre-shading march record bit for bit, RGBA within
the north-star tolerance of 1e-4."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RGBA_TOL = 1e-4


def load_tool():
    spec = importlib.util.spec_from_file_location("full_parity", os.path.join(ROOT, "tools", "full_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("pipeline", ["plain", "fused", "fused_ilv"])
@pytest.mark.parametrize("side", [256, 512])
def test_whole_grid_and_whole_frame(pkg, oracle, side, pipeline):
    """All three pipelines bench.py times, with the kernel instantiations it times (default options): plain = sdfv_fill_grid +
    the no-aux march over tex0.r; fused = sdfv_fill_grid_commit (nt texture stores, distance volume in the same launch) +
    the no-aux hand-written march over that volume in box-first order; fused_ilv = the fill that writes the march's
    y-interleaved volume itself (256^3: the LDS pair form, 512^3: fill_dense_pairrows_kernel -- what SDFViewer runs beyond
    the last-level cache) + the march over that volume.  Textures (and the volume) word for word, the
    no-aux RGBA on every pixel against the oracle and bit for bit against the aux kernel's."""
    bad_words, aux_diff, rgba_err = load_tool().check(side, log=lambda m: None, pipeline=pipeline)
    assert bad_words == 0
    assert all(v == 0 for v in aux_diff.values()), aux_diff
    assert rgba_err <= RGBA_TOL


def test_config4_grid_as_eight_slabs_on_one_gpu(pkg, oracle):
    """1024^3 (68.7 GB of textures) filled slab by slab as the 8 ranks of config 4 would, every voxel against the oracle."""
    import torch
    if torch.cuda.mem_get_info()[0] < 80 << 30:
        pytest.skip("needs 80 GB of free HBM")
    assert load_tool().check_config4(log=lambda m: None) == 0


def test_config4_slabs_through_the_fused_step_in_rccl_loopback(pkg, oracle):
    """Config 4's eight slabs of 1024^3, each through sdfv_slab_fill_step_commit (what bench.py --gpus N times per rank) on
    the library's RCCL communicator in loopback, default options: textures, distance volume and ghosts."""
    import torch
    if torch.cuda.mem_get_info()[0] < 24 << 30:
        pytest.skip("needs 24 GB of free HBM")
    assert load_tool().check_config4_fused(log=lambda m: None) == 0


def test_config5_batch_of_64_cameras(pkg, oracle):
    """BASELINE.json config 5's shape on one GPU: 64 orbit cameras x 1080p over the 256^3 grid in ONE call; every
    frame within tolerance of the oracle's, four of them also bit for bit before shading."""
    import numpy as np
    import torch
    W, H, n = 1920, 1080, 64
    prm = pkg.default_params()
    g = pkg.make_grid((256, 256, 256))
    t0, t1 = pkg.alloc_textures(g)
    dist = torch.empty((256, 256, 256), dtype=torch.float32, device=t0.device)
    pkg.fill_grid(prm, g, t0, t1, dist=dist)  # the fused fill: the volume bench.py's batch marches over
    rp = pkg.default_render_params(g)
    cams = pkg.orbit_cameras(n, aspect=W / H)
    rgba = pkg.raymarch(rp, t0, t1, cams, W, H, dist=dist)
    # config 5 as BASELINE names it -- the image-tile split: the 8 ranks' row bands of every camera (split_rows), each
    # band one call like bench.py --batch-split rows makes per rank; assembled they are the whole-image batch bit for bit
    import importlib
    par = importlib.import_module("sdf-viewer_amd.parallel")
    banded = torch.empty_like(rgba)
    for r in range(8):
        y0, y1 = par.split_rows(H, r, 8)
        banded[:, y0:y1] = pkg.raymarch(rp, t0, t1, cams, W, H, y0=y0, y1=y1, dist=dist)
    torch.cuda.synchronize()
    assert torch.equal(banded.view(torch.int32), rgba.view(torch.int32))
    del banded
    h0, h1 = t0.cpu().numpy(), t1.cpu().numpy()
    orp = oracle.copy_struct(oracle.RenderParams, rp)
    worst = 0.0
    for k in range(n):
        want, want_aux = oracle.raymarch(orp, h0, h1, oracle.copy_struct(oracle.Camera, cams[k]), W, H, threads=16,
                                         want_aux=(k % 16 == 0))
        worst = max(worst, float(np.abs(rgba[k].cpu().numpy() - want).max()))
        if k % 16 == 0:
            _, aux = pkg.raymarch(rp, t0, t1, cams[k], W, H, want_aux=True, dist=dist)
            got = aux[0].cpu().numpy().view(oracle.AUX_DTYPE).reshape(H, W)
            for f in ("status", "steps", "hit_pos", "t", "raw0", "raw1", "normal", "depth"):
                assert (got[f].view(np.uint32) == want_aux[f].view(np.uint32)).all(), (k, f)
    assert worst <= RGBA_TOL


def test_config4_sharded_march_equals_single_gpu_march(pkg):
    """Config 4's 1024^3 grid held as 8 z-slabs (+ ghosts) next to the whole grid on ONE GPU (138 GB): the sharded
    march, ranks run in lockstep, against sdfv_raymarch over the whole grid -- every pixel and aux word (normals apart)."""
    import importlib
    import numpy as np
    import torch
    if torch.cuda.mem_get_info()[0] < 170 << 30:
        pytest.skip("needs 170 GB of free HBM")
    par = importlib.import_module("sdf-viewer_amd.parallel")
    spec = importlib.util.spec_from_file_location("sharded", os.path.join(ROOT, "tests", "test_gpu_sharded_march.py"))
    sharded = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharded)
    dims, world, W, H = (1024, 1024, 1024), 8, 1920, 1080
    bb = ((-1, -1, -1), (1, 1, 1))
    prm = pkg.default_params()
    full = pkg.make_grid(dims)
    f0, f1 = pkg.alloc_textures(full)
    pkg.fill_grid(prm, full, f0, f1)
    rp = pkg.default_render_params(full)
    cam = pkg.camera_look_at(aspect=W / H)
    want_rgba, want_aux = pkg.raymarch(rp, f0, f1, cam, W, H, want_aux=True)
    slabs, grids = sharded.build_slabs(pkg, par, prm, dims, world, bb)
    got_rgba, got_aux, handed = sharded.run_lockstep(pkg, par, rp, slabs, grids, cam, W, H)
    assert torch.equal(got_rgba.view(torch.int32), want_rgba[0].view(torch.int32))
    ga, wa = got_aux.cpu().numpy(), want_aux[0].cpu().numpy()
    np.testing.assert_array_equal(ga[..., :14], wa[..., :14])
    np.testing.assert_array_equal(ga[..., 17], wa[..., 17])
    assert handed > 10000 and (wa[..., 0] == 1).sum() > 100000
