#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --save-temps) for a scratch reload within eight instructions before a global_load_lds: such a reload
is followed by `s_waitcnt vmcnt(0)`, so every LDS-DMA instruction of a gather step waits for the one before it (profiles/
r05_packed_forward.txt, block 8).      python tools/dma_reload_check.py <unit>-hip-amdgcn-amd-amdhsa-gfx950.s ..."""
import re,sys
for f in sys.argv[1:]:
    lines=open(f).read().split('\n')
    cur=None; res={}
    for i,l in enumerate(lines):
        m=re.match(r'^(_Z\w+):',l)
        if m: cur=m.group(1); res[cur]=[0,0,0]; continue
        if cur is None: continue
        t=l.strip().split()
        if not t: continue
        if t[0].startswith('global_load_lds'):
            res[cur][0]+=1
            # scratch reload + vmcnt(0) within the previous 8 instructions?
            window=' '.join(lines[max(0,i-8):i])
            if 'scratch_load' in window: res[cur][1]+=1
        if t[0].startswith('scratch_load'): res[cur][2]+=1
    for k,v in res.items():
        if v[0]: print(f"{k[:75]:75s} dma {v[0]:4d}  dma-after-reload {v[1]:4d}  reloads {v[2]:4d}")
