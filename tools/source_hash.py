#!/usr/bin/env python3
"""The build id of a SOURCE TREE: sha256 over the kernel sources in the order csrc/Makefile hashes them (SRCS, then HDRS),
first 12 hex digits -- what sdfv_build_id() of a library built from this tree answers.  usage: python tools/source_hash.py"""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdf-viewer_amd", "csrc")


def source_hash():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    names = re.search(r"^SRCS\s*:=\s*(.*)$", mk, re.M).group(1).split() + re.search(r"^HDRS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(CSRC, n), "rb").read())
    return h.hexdigest()[:12]


if __name__ == "__main__":
    print(source_hash())
