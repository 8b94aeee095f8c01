#!/usr/bin/env python3
"""The interactive loop of the reference (app/mod.rs: a parameter slider moves -> set_parameter -> changed() ->
SDFViewer::update refills the changed box over three passes -> commit -> the next frame) on the C++ host mirror:
wall time from set_parameter to the finished 1080p frame, frame left on the device."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import host_binding as host  # noqa: E402

W, H = 1920, 1080
for side in (256, 512):
    sdf = host.SDF.demo()
    viewer = host.Viewer.from_bb(sdf.bounding_box(), side, 2)
    while viewer.update(sdf, 0.03) > 0:
        pass
    viewer.commit()
    img = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    viewer.render_device(W, H, img.data_ptr())
    viewer.sync()
    sphere = sdf.children()[1]
    lat = []
    for k in range(40):
        radius = 1.05 - 0.002 * (k % 10 + 1)   # the slider moves a little: a thin shell around the sphere changes
        t0 = time.perf_counter()
        err = sphere.set_parameter(1, float(radius))  # SDFDemoSphere::ID_RADIUS
        updates = 0
        while True:
            n = viewer.update(sdf, 0.03)
            if n == 0:
                break
            updates += n
        viewer.commit()
        viewer.render_device(W, H, img.data_ptr())
        viewer.sync()
        lat.append((time.perf_counter() - t0) * 1e3)
        assert err is None and updates > 0, (err, updates)
    lat.sort()
    print(f"{side}^3 + {W}x{H}: parameter edit -> refill of the changed box (the demo reports its whole box: dense refill) "
          f"-> commit -> frame: "
          f"median {lat[len(lat) // 2]:.3f} ms, best {lat[0]:.3f} ms (host wall clock, {len(lat)} edits)")
