#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels: python tools/asm_stats.py [file.hip] [filter]
Splits each kernel at basic-block labels so the Jacobi sweep loop body can be read separately.

`kernel_flop_mix(asm_path)` (imported by __graft_entry__.build_hip on the assembly the build saved anyway) returns
{demangled kernel name: {"valu": VALU instructions, "flops": fp64 flops, ...}} -- the static flops-per-VALU-instruction
ratio bench.py multiplies the MEASURED VALU instructions per wave with (FMA = 2; add / mul / min / max / rcp / rsq / sqrt /
ldexp / frexp / trig-free fp64 arithmetic = 1; moves, compares, conversions, integer and 32-bit work = 0)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

_F64_TWO = re.compile(r"^v_(fma|fmac|pk_fma)_f64")
_F64_ONE = re.compile(r"^v_(add|mul|min|max|rcp|rsq|sqrt|ldexp|fract|trunc|floor|ceil|rndne|div_scale|div_fmas|div_fixup|pk_add|pk_mul)_f64")


def _demangle(names):
    tool = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not os.path.exists(tool):
        tool = "c++filt"
    try:
        out = subprocess.run([tool], input="\n".join(names).encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def kernel_flop_mix(asm_path):
    lines = open(asm_path).read().split("\n")
    kernels = set()
    for l in lines:                                   # .amdhsa_kernel <symbol> marks the entry points
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            kernels.add(m.group(1))
    stats = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):", lines[i])
        if m and m.group(1) in kernels:
            name = m.group(1)
            valu = flops = f64 = dpp = 0
            i += 1
            while i < len(lines) and "s_endpgm" not in lines[i] and not lines[i].startswith(".Lfunc_end"):
                l = lines[i]
                if l.startswith("\t") and not l.strip().startswith((".", ";")):
                    op = l.split()[0]
                    if op.startswith("v_") and not op.startswith(("v_accvgpr", "v_readlane", "v_readfirstlane", "v_nop")):
                        valu += 1
                        if "f64" in op:
                            f64 += 1
                        if _F64_TWO.match(op):
                            flops += 2
                        elif _F64_ONE.match(op):
                            flops += 1
                        if "dpp" in op or " row_" in l or "quad_perm" in l:
                            dpp += 1
                i += 1
            stats[name] = {"valu": valu, "flops": flops, "f64_instructions": f64, "dpp": dpp}
        i += 1
    pretty = _demangle(list(stats))
    def short(sym):
        sym = sym.replace("(anonymous namespace)::", "").replace("void ", "", 1).strip()
        depth = 0
        for i, ch in enumerate(sym):          # cut at the parameter list: the first "(" outside the template arguments
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                return sym[:i].strip()
        return sym
    return {short(pretty[k]): v for k, v in stats.items()}


def main():
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


if __name__ == "__main__":
    main()
