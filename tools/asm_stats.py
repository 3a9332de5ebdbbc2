#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels: python tools/asm_stats.py [file.hip] [filter]
Splits each kernel at basic-block labels so the Jacobi sweep loop body can be read separately."""
import collections
import re
import subprocess
import sys
import tempfile

src = sys.argv[1] if len(sys.argv) > 1 else "sympa_amd/csrc/siegel_dist.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else "ILi4ELi0E"
extra = sys.argv[3:] 
with tempfile.TemporaryDirectory() as d:
    out = f"{d}/k.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           "-o", out, src] + extra, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\S+):", lines[i])
    if m and flt in m.group(1) and ("siegel" in m.group(1) or "allpairs" in m.group(1)):
        name = m.group(1)
        blocks = collections.OrderedDict()
        cur = "entry"
        blocks[cur] = []
        i += 1
        while i < len(lines) and "s_endpgm" not in lines[i]:
            l = lines[i]
            mb = re.match(r"^(\.LBB\S+):", l)
            if mb:
                cur = mb.group(1)
                blocks[cur] = []
            elif l.startswith("\t") and not l.strip().startswith((".", ";")):
                blocks[cur].append(l.split()[0])
            i += 1
        print(name)
        tot = collections.Counter()
        for b, ins in blocks.items():
            c = collections.Counter(ins)
            tot.update(c)
            f64 = sum(v for k, v in c.items() if "f64" in k)
            trans = sum(v for k, v in c.items() if re.search(r"v_(rcp|rsq|sqrt|log|exp|div_scale|div_fmas|div_fixup)", k))
            print(f"  {b:14s} n={len(ins):5d} f64={f64:5d} trans/div={trans:4d} mem={sum(v for k,v in c.items() if k.startswith(('global_','buffer_','scratch_','ds_','flat_'))):4d}")
        print("  TOTAL", sum(tot.values()), tot.most_common(30))
    i += 1
