#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of the kernels whose mangled name contains ARG1, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks over the library's translation units (the library's own flags; extra flags
after ARG1).  Usage: python tools/kernel_resources.py mlp_forward [-DPNR_MLP_RING=7]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pioneer_amd import _lib  # noqa: E402

for unit in _lib.UNITS:
    cmd = ["/opt/rocm/bin/hipcc", *_lib.HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(_lib.CSRC, unit),
           "-o", "/tmp/pnr_kernel_resources.o"] + sys.argv[2:]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    txt = r.stderr
    if r.returncode:
        print(txt[-3000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", txt)
    for b in blocks[1:]:
        name = b.split("\n")[0].strip()
        if sys.argv[1] in name:
            g = lambda pat: (re.search(pat, b) or [None, None])[1]  # noqa: E731
            print(name[:70], "VGPR", g(r"VGPRs: (\d+)"), "AGPR", g(r"AGPRs: (\d+)"), "SGPR", g(r"SGPRs: (\d+)"), "scratch", g(r"ScratchSize \[bytes/lane\]: (\d+)"),
                  "occ", g(r"Occupancy \[waves/SIMD\]: (\d+)"), "lds", g(r"LDS Size \[bytes/block\]: (\d+)"))
