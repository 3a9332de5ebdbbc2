#!/usr/bin/env python3
"""Static check of the gfx950 ISA for the hazard the compiler cannot see in inline asm (sympa_amd/csrc/spd_coop.hpp):
a VGPR that a DPP instruction reads as its DPP source (src0) must not be written by a VALU instruction in the two
preceding wait states.  Walks every kernel of an assembly file (hipcc -S --cuda-device-only, or the .s that
--save-temps leaves), over ALL predecessors of a basic block.  A DPP instruction with a partial row_mask / bank_mask also reads its DESTINATION (the lanes it masks off keep the old value)
under the same rule.  Also flags v_cmpx (a VALU write of EXEC needs five wait
states before a DPP instruction; gfx950 code normally has none).

    python tools/check_dpp_hazards.py file.s [...]        exit code 1 when a hazard is found
"""
import re
import sys

WAIT_STATES = 2


def vregs(tok):
    tok = tok.strip().rstrip(",").lstrip("-|").rstrip("|")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


def kernels(lines):
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            j = i + 1
            while j < len(lines) and not lines[j].startswith("\t.end_amdhsa_kernel") and "s_endpgm" not in lines[j]:
                j += 1
            yield m.group(1), lines[i + 1:j + 1]
            i = j
        i += 1


def check_kernel(name, text):
    # basic blocks: label -> list of instructions
    blocks, order, cur = {"entry": []}, ["entry"], "entry"
    for l in text:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            ins = l.split(";")[0].strip()
            if ins:
                blocks[cur].append(ins)
    preds = {b: set() for b in order}
    for k, b in enumerate(order):
        ins = blocks[b]
        falls = True
        for x in ins:
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", x)
            if m and m.group(1) in preds:
                preds[m.group(1)].add(b)
        if ins and ins[-1].startswith("s_branch"):
            falls = False
        if ins and ins[-1].startswith("s_endpgm"):
            falls = False
        if falls and k + 1 < len(order):
            preds[order[k + 1]].add(b)

    found = []

    def walk(block, idx, src, ws, seen, consumer):
        """look backwards from instruction idx (exclusive) of `block` with `ws` wait states already passed"""
        ins = blocks[block]
        j = idx - 1
        while j >= 0 and ws < WAIT_STATES:
            p = ins[j]
            if p.startswith("s_nop"):
                ws += int(p.split()[1], 0) + 1
            else:
                ops = p.split(None, 1)
                if p.startswith("v_") and len(ops) > 1:
                    dst = vregs(ops[1].split(",")[0])
                    if dst & src:
                        found.append((name, p, consumer))
                        return
                ws += 1
            j -= 1
        if ws < WAIT_STATES and j < 0:
            for pb in preds[block]:
                key = (pb, ws)
                if key not in seen:
                    seen.add(key)
                    walk(pb, len(blocks[pb]), src, ws, seen, consumer)

    ndpp = 0
    for b in order:
        for i, l in enumerate(blocks[b]):
            if l.startswith("v_cmpx"):
                found.append((name, l, "VALU write of EXEC"))
            if "_dpp" in l.split()[0]:
                ndpp += 1
                ops = l.split(None, 1)[1].split(",")
                src = vregs(ops[1].split()[0])
                # a lane masked off by row_mask / bank_mask keeps the OLD value of the destination, and that value is read
                # like a DPP source (measured: tools/microbench/dpp_bank_mask.hip) -- the destination counts as a source then
                m = re.search(r"row_mask:(0x[0-9a-fA-F]+)\s+bank_mask:(0x[0-9a-fA-F]+)", l)
                if m and (int(m.group(1), 16) != 0xf or int(m.group(2), 16) != 0xf):
                    src = src | vregs(ops[0])
                walk(b, i, src, 0, set(), l)
    return ndpp, found


def check_file(path):
    lines = open(path).read().split("\n")
    total, bad = 0, []
    for name, text in kernels(lines):
        n, f = check_kernel(name, text)
        total += n
        bad += f
    return total, bad


if __name__ == "__main__":
    rc = 0
    for path in sys.argv[1:]:
        total, bad = check_file(path)
        print(f"{path}: {total} DPP instructions, {len(bad)} hazards")
        for name, prod, cons in bad[:20]:
            print(f"  {name}\n    {prod}\n    {cons}")
        rc |= 1 if bad else 0
    sys.exit(rc)
