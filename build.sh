#!/bin/bash
# Build the HIP library (gfx950) and the CPU oracle in-tree.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE/pioneer_amd/csrc"
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
    -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function \
    "$@" -o libpioneer_amd.so pnr_api.hip
make -s -C "$HERE/oracle"
echo "built $HERE/pioneer_amd/csrc/libpioneer_amd.so"
