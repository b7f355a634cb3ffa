#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of the kernels whose mangled name contains ARG1, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks over pioneer_amd/csrc/pnr_api.hip (the library's own flags; extra flags
after ARG1).  Usage: python tools/kernel_resources.py mlp_forward [-DPNR_MLP_RING=7]"""
import subprocess, re, sys
hipcc="/opt/rocm/bin/hipcc"
cmd=[hipcc,"-O3","--offload-arch=gfx950","-std=c++17","-fPIC","-shared","-ffp-contract=off","-fno-slp-vectorize","-mllvm","-amdgpu-kernarg-preload-count=16","-Rpass-analysis=kernel-resource-usage","pioneer_amd/csrc/pnr_api.hip","-Iinclude","-o","/tmp/pnr_kernel_resources.so"]+sys.argv[2:]
r=subprocess.run(cmd,capture_output=True,text=True,cwd="/root/repo")
txt=r.stderr
if r.returncode: print(txt[-3000:])
blocks=re.split(r"remark: [^\n]*Function Name: ", txt)
for b in blocks[1:]:
    name=b.split("\n")[0].strip()
    if sys.argv[1] in name:
        vg=re.search(r"VGPRs: (\d+)",b); ag=re.search(r"AGPRs: (\d+)",b); sc=re.search(r"ScratchSize \[bytes/lane\]: (\d+)",b); occ=re.search(r"Occupancy \[waves/SIMD\]: (\d+)",b); lds=re.search(r"LDS Size \[bytes/block\]: (\d+)",b)
        print(name[:60], "VGPR",vg and vg.group(1),"AGPR",ag and ag.group(1),"scratch",sc and sc.group(1),"occ",occ and occ.group(1),"lds",lds and lds.group(1))
