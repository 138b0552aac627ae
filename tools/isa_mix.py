#!/usr/bin/env python3
"""Instruction mix of the MFMA-carrying basic blocks of one kernel symbol (hipcc -S output).
   python tools/isa_mix.py <file.hip> <mangled-substring>"""
import re
import subprocess
import sys

src, key = sys.argv[1], sys.argv[2]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                       src, "-o", "/tmp/_mix.s"], stderr=subprocess.DEVNULL)
s = open("/tmp/_mix.s").read()
names = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if key in m]
for name in names[:4]:
    i = s.index(name + ":")
    j = s.index(".end_amdhsa_kernel", i)
    cur, blocks = None, {}
    for ln in s[i:j].split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        t = ln.strip()
        if m:
            cur = m.group(1)
            blocks[cur] = []
        elif cur and t and not t.startswith((";", ".")):
            blocks[cur].append(t)
    print(name)
    for b, ins in blocks.items():
        if not any(x.startswith("v_mfma") for x in ins) and len(ins) < 40:
            continue
        cnt = {}
        for x in ins:
            op = x.split()[0]
            k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else
                 "wait" if op.startswith("s_waitcnt") else "barrier" if op.startswith("s_barrier") else
                 "salu" if op.startswith("s_") else "vmem" if op.startswith(("buffer_", "global_", "flat_")) else
                 "lds" if op.startswith("ds_") else op)
            cnt[k] = cnt.get(k, 0) + 1
        print("  ", b, len(ins), cnt)
