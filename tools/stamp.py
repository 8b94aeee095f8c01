#!/usr/bin/env python3
"""One header line for a profile summary or a JSON of evidence: which build of libsdfgrid.so and which box it describes --
"# build_id: <sdfv_build_id()>  box: <GPU unique id>" (the format tools/pmc_to_traffic.py reads back).  --json: {"build_id", "box"}."""
import json
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_common import running_build_id  # noqa: E402


def box_uuid():
    try:
        out = subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=30).stdout
        m = re.search(r"Unique ID:\s*(\S+)", out)
        return m.group(1) if m else None
    except Exception:
        return None


if __name__ == "__main__":
    if "--json" in sys.argv:
        print(json.dumps({"build_id": running_build_id(), "box": box_uuid()}))
    else:
        print(f"# build_id: {running_build_id()}  box: {box_uuid()}")
