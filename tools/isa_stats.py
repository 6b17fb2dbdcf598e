#!/usr/bin/env python3
"""Static instruction mix of the stream_collide kernels in a device assembly file (hipcc --cuda-device-only -S): VALU / packed / SALU / DS / global
instructions, registers, LDS and scratch per kernel.   usage: tools/isa_stats.py <file.s> [name filter regex]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile("k_stream_collide")
ks, name = {}, None
for line in txt.splitlines():
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1); ks[name] = []; continue
    if name and line.startswith(".Lfunc_end"):
        name = None; continue
    if name:
        t = line.strip()
        if t and not t.startswith((";", ".")):
            ks[name].append(t)
meta = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    g = lambda k: int((re.search(k + r" (\d+)", m.group(2)) or [0, 0])[1])
    meta[m.group(1)] = (g("amdhsa_next_free_vgpr"), g("amdhsa_accum_offset"), g("amdhsa_next_free_sgpr"), g("amdhsa_group_segment_fixed_size"),
        g("amdhsa_private_segment_fixed_size"))
print("%-92s %5s %5s %5s %5s %4s %4s %4s  vgpr/accoff/sgpr/lds/scratch" % ("kernel", "VALU", "pk", "mov", "SALU", "DS", "GLB", "acc"))
for n, b in ks.items():
    if not flt.search(n):
        continue
    c = lambda p: sum(1 for t in b if t.startswith(p))
    print("%-92s %5d %5d %5d %5d %4d %4d %4d  %s" % (n[:92], c("v_"), c("v_pk_"), c("v_mov") + c("v_pk_mov"), c("s_"), c("ds_"), c("global_"),
        c("v_accvgpr"), meta.get(n)))
