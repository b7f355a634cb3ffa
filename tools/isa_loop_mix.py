#!/usr/bin/env python3
"""Static instruction mix of the largest loop of a kernel in an ISA listing.  First:
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 \\
        -Iinclude --cuda-device-only -S pioneer_amd/csrc/pnr_api.hip -o /tmp/pnr_api.s
then: python tools/isa_loop_mix.py MANGLED_PREFIX [TOP_N] (e.g. _ZN3pnr15dyn_step_kernelILb1ELb1ELb1ELi0E: the sub-step loop)."""
import re, collections, sys
lines=open('/tmp/pnr_api.s').read().split("\n")
name=sys.argv[1]
start=[i for i,l in enumerate(lines) if l.startswith(name) and ": " in l][0]
end=next(i for i in range(start,len(lines)) if "s_endpgm" in lines[i])
body=lines[start+1:end]
def count(seg):
    ops=collections.Counter()
    for line in seg:
        line=line.strip()
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"): continue
        ops[line.split()[0]]+=1
    return ops
labs={l.strip().split(":")[0]:i for i,l in enumerate(body) if re.match(r"\s*\.LBB\d+_\d+:",l)}
best=None
for i,l in enumerate(body):
    m=re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)",l)
    if m and m.group(1) in labs and labs[m.group(1)]<i:
        if best is None or i-labs[m.group(1)]>best[1]-best[0]: best=(labs[m.group(1)],i)
lo=count(body[best[0]:best[1]])
valu=sum(c for o,c in lo.items() if o.startswith("v_")); pk=sum(c for o,c in lo.items() if o.startswith("v_pk"))
print("total static",sum(count(body).values()),"loop",sum(lo.values()),"valu",valu,"packed",pk)
for op,c in lo.most_common(int(sys.argv[2]) if len(sys.argv)>2 else 12): print(f"  {op:26s} {c}")
