#!/bin/bash
# Build the HIP library (gfx950; two translation units compiled in parallel, see pioneer_amd/_lib.py) and the CPU oracle in-tree.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE"
python3 -c 'from pioneer_amd import _lib; print("built", _lib.build_library(force=True, verbose=True))'
make -s -C "$HERE/oracle"
