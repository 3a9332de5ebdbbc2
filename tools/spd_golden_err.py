#!/usr/bin/env python3
"""Relative error of the spd forward kernels against the 50-digit goldens, per case: python tools/spd_golden_err.py [n]
(with SYMPA_HIP_LIB=<variant.so> for a build of tools/build_variant.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sympa_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"spd_n{n}.npz"))
for case in g["case_names"]:
    x, y, want = torch.from_numpy(g[f"{case}__x"]), torch.from_numpy(g[f"{case}__y"]), g[f"{case}__dist_exact50"]
    errs = []
    for flags in (0, ops.FLAG_GENERIC):
        got = ops.spd_dist_forward(x.to(dev), y.to(dev), flags=flags).cpu().numpy()
        errs.append(float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-13))))
    print(f"n={n} {str(case):10s} lanes-per-pair {errs[0]:.2e}   one-lane {errs[1]:.2e}")
