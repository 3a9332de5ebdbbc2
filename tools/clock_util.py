"""Host side of C-ABI sympa_clock_stamp: pair the stamps two launches left on the SAME CU and turn them into a clock."""
BLOCKS = 2048


def stamp(lib, dev, torch):
    buf = torch.zeros(3 * BLOCKS, dtype=torch.int64, device=dev)
    rc = lib.sympa_clock_stamp(buf.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    return buf if rc == 0 else None


def _by_cu(buf):
    """{(xcc, se/sh/cu bits of HW_ID): (s_memtime, s_memrealtime)} -- of the blocks that ran on a CU the earliest one."""
    rows = buf.cpu().view(BLOCKS, 3).tolist()
    out = {}
    for t, r, where in rows:
        key = (where & 0xF, (where >> 16) & 0xFF)          # XCC_ID, HW_ID[15:8] = CU, SH, SE
        if key not in out or r < out[key][1]:
            out[key] = (t, r)
    return out


def between(s0, s1):
    """Clock over the region between two stamps: per CU d(s_memtime) / d(s_memrealtime) x 100 MHz; median / min / max over the CUs
    that appear in both stamps, and the median per XCD."""
    if s0 is None or s1 is None:
        return None
    a, b = _by_cu(s0), _by_cu(s1)
    per_cu, per_xcd = [], {}
    us = []
    for key in a.keys() & b.keys():
        dt, dr = b[key][0] - a[key][0], b[key][1] - a[key][1]
        if dr > 0 and dt > 0:
            mhz = dt / dr * 100.0
            per_cu.append(mhz)
            per_xcd.setdefault(key[0], []).append(mhz)
            us.append(dr / 100.0)
    if not per_cu:
        return None
    per_cu.sort()
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    return {"mhz": med(per_cu), "mhz_min_cu": per_cu[0], "mhz_max_cu": per_cu[-1], "cus_paired": len(per_cu),
            "mhz_per_xcd": {str(x): med(v) for x, v in sorted(per_xcd.items())}, "region_us": med(us)}
