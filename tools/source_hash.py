#!/usr/bin/env python3
"""Which build is this?  Two answers, one tool.
(no argument)  the build id of the SOURCE TREE: sha256 over the kernel sources in the order csrc/Makefile hashes them (SRCS, then
               HDRS), first 12 hex digits -- what sdfv_build_id() of a library built from this tree answers (no library, no GPU).
--stamp        one header line for a profile summary: which build of libsdfgrid.so is LOADED and which box it runs on --
               "# build_id: <sdfv_build_id()>  box: <GPU unique id>" (the format tools/pmc_to_traffic.py reads back).
--json         the same as {"build_id", "box"}."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdf-viewer_amd", "csrc")


def source_hash():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    names = re.search(r"^SRCS\s*:=\s*(.*)$", mk, re.M).group(1).split() + re.search(r"^HDRS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(CSRC, n), "rb").read())
    return h.hexdigest()[:12]


def box_uuid():
    try:
        out = subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=30).stdout
        m = re.search(r"Unique ID:\s*(\S+)", out)
        return m.group(1) if m else None
    except Exception:
        return None


if __name__ == "__main__":
    if "--stamp" in sys.argv or "--json" in sys.argv:
        sys.path.insert(0, ROOT)
        from bench_common import running_build_id
        if "--json" in sys.argv:
            print(json.dumps({"build_id": running_build_id(), "box": box_uuid()}))
        else:
            print(f"# build_id: {running_build_id()}  box: {box_uuid()}")
    else:
        print(source_hash())
