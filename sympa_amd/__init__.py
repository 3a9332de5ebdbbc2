"""sympa_amd -- MI355X-native Siegel-space distance hot path behind the reference's Python API.

Only what the path `Model.forward -> manifold.dist` needs lives here:
  csrc/       hand-written HIP kernels for gfx950 + the C-ABI (include/sympa_hip.h)
  _lib.py     ctypes binding of libsympa_hip.so (fails loudly when the library is missing)
  ops.py      torch-facing entry points (device pointers + current HIP stream -> C-ABI)
  manifolds/  UpperHalfManifold / BoundedDomainManifold / metrics with the reference signatures
  model.py, embeddings.py, losses.py   the callers on either side of the path
"""
from sympa_amd import config  # noqa: F401  (sets the fp64 default dtype like the reference)

__version__ = "0.1.0"
