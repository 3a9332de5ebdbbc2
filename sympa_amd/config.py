"""Constants of the hot path (reference sympa/config.py:17-21).  The reference sets the global
default dtype to float64 at import time (config.py:17-18); so does this module, because the
drop-in contract includes fp64 tables and fp64 distances."""
import torch

DEFAULT_DTYPE = torch.float64
torch.set_default_dtype(DEFAULT_DTYPE)
EPS = {torch.float32: 4e-3, torch.float64: 1e-5}

INIT_EPS = 1e-3
BURNIN_FACTOR = 10
