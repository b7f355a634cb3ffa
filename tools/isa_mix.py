#!/usr/bin/env python3
"""Static instruction mix of kernels in an ISA listing of pnr_learn.hip (every instruction counted once: the MLP kernels are fully
unrolled, so static ~ dynamic per wave and tile).  First:
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 \\
        -Iinclude --cuda-device-only -S pioneer_amd/csrc/pnr_learn.hip -o /tmp/pnr_learn.s [-DVARIANT..]
then: python tools/isa_mix.py /tmp/pnr_learn.s SUBSTRING [SUBSTRING ...]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")


def kernel(sub):
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sub in l.split(":")[0]][0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start].split(":")[0], lines[start + 1:end]


CATS = [("mfma", ("v_mfma",)), ("trans", ("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")), ("v_pk", ("v_pk",)), ("cvt", ("v_cvt",)),
        ("valu", ("v_",)), ("lds", ("ds_",)), ("vmem", ("global_", "buffer_", "flat_", "scratch_")), ("waitcnt", ("s_waitcnt",)), ("barrier", ("s_barrier",)),
        ("salu", ("s_",))]
for sub in sys.argv[2:]:
    name, body = kernel(sub)
    ops = collections.Counter()
    for line in body:
        line = line.strip()
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    cat = collections.Counter()
    for o, c in ops.items():
        cat[next((n for n, pre in CATS if o.startswith(pre)), "other")] += c
    print(name[:60], sum(ops.values()), dict(cat))
    print("   ", ", ".join(f"{o} {c}" for o, c in ops.most_common(30)))
