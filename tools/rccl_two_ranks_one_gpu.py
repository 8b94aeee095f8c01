#!/usr/bin/env python3
"""Two PROCESSES on ONE GPU over the library's own RCCL communicator (non-periodic world of 2): does RCCL accept two
ranks on the same device?  If it does, the z-slab fill step runs with a real rank-to-rank exchange and the ghosts are
checked; if it refuses, the error string is reported.  The 128-byte id travels through a file (no torch.distributed).
Parent: python tools/rccl_two_ranks_one_gpu.py           -> one JSON line
Child:  python tools/rccl_two_ranks_one_gpu.py <rank> <id file>"""
import ctypes as C
import importlib
import re
import json
import os
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rank, id_path):
    import torch
    pkg = importlib.import_module("sdf-viewer_amd")
    par = importlib.import_module("sdf-viewer_amd.parallel")
    K = pkg._capi
    torch.cuda.set_device(0)
    ident = (C.c_ubyte * K.COMM_ID_BYTES)()
    if rank == 0:
        rc = pkg.lib.sdfv_slab_comm_unique_id(ident)
        if rc:
            print(json.dumps({"rank": rank, "stage": "unique_id", "rc": rc, "error": pkg.lib.sdfv_last_error().decode()}))
            return
        with open(id_path + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.rename(id_path + ".tmp", id_path)
    else:
        t_end = time.time() + 60
        while not os.path.exists(id_path) and time.time() < t_end:
            time.sleep(0.05)
        ident = (C.c_ubyte * K.COMM_ID_BYTES)(*open(id_path, "rb").read())
    handle = C.c_void_p()
    rc = pkg.lib.sdfv_slab_comm_create(ident, rank, 2, 0, C.byref(handle))
    if rc:
        print(json.dumps({"rank": rank, "stage": "comm_create", "rc": rc, "error": pkg.lib.sdfv_last_error().decode()}))
        return
    dims = (64, 64, 32)
    prm = pkg.default_params()
    slab = par.alloc_slab(dims, rank, 2, "cuda", fill_value=-7.0)
    grid = pkg.make_grid(dims, z_begin=slab.z_begin, z_end=slab.z_end)
    rc = pkg.lib.sdfv_slab_fill_step(handle, C.byref(prm), 0, C.byref(grid), C.c_void_p(slab.tex0.data_ptr()),
                                     C.c_void_p(slab.tex1.data_ptr()), None)
    torch.cuda.synchronize()
    full0, full1 = pkg.alloc_textures(pkg.make_grid(dims))
    pkg.fill_grid(prm, pkg.make_grid(dims), full0, full1)
    torch.cuda.synchronize()
    lo, hi = slab.z_begin - slab.ghost_lo, slab.z_end + slab.ghost_hi
    ok = bool(rc == 0 and torch.equal(slab.tex0, full0[lo:hi]) and torch.equal(slab.tex1, full1[lo:hi]))
    pkg.lib.sdfv_slab_comm_destroy(handle)
    print(json.dumps({"rank": rank, "stage": "fill_step", "rc": rc, "ghosts_and_owned_equal_dense_fill": ok}))


def parent():
    with tempfile.TemporaryDirectory() as d:
        id_path = os.path.join(d, "id")
        env = dict(os.environ, NCCL_DEBUG="WARN")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), id_path], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, env=env) for r in (0, 1)]
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                p.kill()
                o, e = p.communicate()
                o += json.dumps({"stage": "timeout"})
            found = re.findall(r'\{"rank": \d.*?\}', o)  # RCCL's own WARN lines share stdout and may interleave
            warn = sorted({m for m in re.findall(r"NCCL WARN ([^\[\n{]*)", o + e) if "Could not read node" not in m})
            outs.append({"result": json.loads(found[-1]) if found else None, "rccl_messages": warn[-3:]})
    accepted = all(o["result"] and o["result"].get("stage") == "fill_step" for o in outs)
    print(json.dumps({"backend": "rccl", "n_gpus": 1, "ranks": 2, "same_device_accepted": accepted, "ranks_out": outs}))


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        child(int(sys.argv[1]), sys.argv[2])
    else:
        parent()
