#!/usr/bin/env python3
"""Opcode histogram of one kernel of a device assembly file.   usage: tools/isa_hist.py <file.s> <kernel name regex> [top N]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for m in re.finditer(r"^(_Z\w+):", txt, re.M):
    if pat.search(m.group(1)):
        j = txt.index(".Lfunc_end", m.end())
        body = [l.strip() for l in txt[m.end():j].splitlines() if l.strip() and not l.strip().startswith((";", "."))]
        c = collections.Counter(l.split()[0] for l in body)
        print(m.group(1)[:100], len(body), "instructions")
        print("  ".join("%d %s" % (v, k) for k, v in c.most_common(top)))
        break
