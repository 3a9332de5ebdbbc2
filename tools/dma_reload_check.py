#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --save-temps) for a scratch reload within eight instructions before a global_load_lds: such a reload
is followed by `s_waitcnt vmcnt(0)`, so every LDS-DMA instruction of a gather step waits for the one before it (found in round 5:
profiles/r05_packed_forward.txt, block 8).  The build runs check_file over every unit (__graft_entry__.build_hip) and writes the
counts into sympa_amd/csrc/dpp_hazard_report.txt; tests/test_abi.py wants them zero.
    python tools/dma_reload_check.py <unit>-hip-amdgcn-amd-amdhsa-gfx950.s ..."""
import re
import sys

WINDOW = 8


def check_file(path):
    """{kernel symbol: (LDS-DMA instructions, of them behind a scratch reload, scratch reloads)} for kernels with DMA."""
    lines = open(path).read().split("\n")
    cur, res = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            res[cur] = [0, 0, 0]
            continue
        if cur is None:
            continue
        t = l.strip().split()
        if not t:
            continue
        if t[0].startswith("global_load_lds"):
            res[cur][0] += 1
            if "scratch_load" in " ".join(lines[max(0, i - WINDOW):i]):
                res[cur][1] += 1
        if t[0].startswith("scratch_load"):
            res[cur][2] += 1
    return {k: tuple(v) for k, v in res.items() if v[0]}


def totals(path):
    r = check_file(path)
    return sum(v[0] for v in r.values()), sum(v[1] for v in r.values())


if __name__ == "__main__":
    for f in sys.argv[1:]:
        for k, v in check_file(f).items():
            print(f"{k[:75]:75s} dma {v[0]:4d}  dma-after-reload {v[1]:4d}  reloads {v[2]:4d}")
