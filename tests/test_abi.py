"""CPU (`-m "not gpu"`): the C-ABI library loads, exports every symbol include/sympa_hip.h declares,
and validates arguments (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from sympa_amd import _lib
from tests.helpers import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sympa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sympa_[a-z_0-9]+)\s*\(", text)))


def test_header_matches_binding():
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for s in declared_symbols():
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.sympa_version()
    assert lib.sympa_max_dims() == 8


def test_argument_validation_without_gpu():
    lib = _lib.load()
    one = ctypes.c_void_p(16)   # never dereferenced: validation happens before any launch
    # b == 0 is a no-op
    assert lib.sympa_siegel_dist_fwd(one, one, 0, 4, 0, 0, None, 1e-5, one, None, None, 0, None) == 0
    assert lib.sympa_siegel_dist_fwd(one, one, -1, 4, 0, 0, None, 1e-5, one, None, None, 0, None) == -1
    assert lib.sympa_siegel_dist_fwd(None, one, 8, 4, 0, 0, None, 1e-5, one, None, None, 0, None) == -1
    assert lib.sympa_siegel_dist_fwd(one, one, 8, 4, 7, 0, None, 1e-5, one, None, None, 0, None) == -1   # model
    assert lib.sympa_siegel_dist_fwd(one, one, 8, 4, 0, 9, None, 1e-5, one, None, None, 0, None) == -1   # metric
    assert lib.sympa_siegel_dist_fwd(one, one, 8, 4, 0, 4, None, 1e-5, one, None, None, 0, None) == -1   # wsum w/o w
    assert lib.sympa_siegel_dist_fwd(one, one, 8, 4, 0, 0, None, 0.0, one, None, None, 0, None) == -1    # eps
    assert lib.sympa_siegel_dist_fwd(one, one, 8, 99, 0, 0, None, 1e-5, one, None, None, 0, None) == -2  # dims
    assert b"dims" in lib.sympa_last_error()
    assert lib.sympa_model_forward(one, 10, 4, None, 2, one, 2, 8, 0, 0, None, 1e-5, None, 1.0, one, None, 0, None) == -1
    assert lib.sympa_model_forward(one, 0, 4, one, 2, one, 2, 8, 0, 0, None, 1e-5, None, 1.0, one, None, 0, None) == -1


def test_digest_and_refresh_entries_validate_their_arguments_without_gpu():
    lib = _lib.load()
    one, al = ctypes.c_void_p(8), ctypes.c_void_p(64)   # never dereferenced
    assert lib.sympa_table_digest(None, 64, al, 0, None) == -1
    assert lib.sympa_table_digest(al, 64, None, 0, None) == -1
    assert lib.sympa_table_digest(al, 0, al, 0, None) == -1
    assert lib.sympa_table_digest(al, 12, al, 0, None) == -1          # not a multiple of 8
    assert lib.sympa_table_digest(one, 64, al, 0, None) == -1         # data not 16-byte aligned
    assert lib.sympa_table_pack_refresh(al, 10, 4, 0, al, 1 << 20, al, 0, None, None) == -2      # dims outside 5..8
    assert lib.sympa_table_pack_refresh(al, 10, 8, 0, al, 1 << 20, None, 0, None, None) == -1    # no digest state
    assert lib.sympa_table_pack_refresh(al, 0, 8, 0, al, 1 << 20, al, 0, None, None) == -1       # empty table
    assert lib.sympa_spd_table_pack_refresh(al, 10, 3, al, 1 << 20, al, 0, None, None) == -2
    assert lib.sympa_spd_table_pack_refresh(al, 10, 16, al, 1 << 20, None, 0, None, None) == -1


def test_merged_rows_flag_is_refused_where_no_kernel_honours_it():
    """SYMPA_FLAG_MERGE_SRC with per-pair rows (round-5 advice): only the one-lane kernels of dims <= 6 write the merged layout; the
    split / eight-lanes / dims >= 7 kernels would write every row while the caller's slot list drops all but the run ends."""
    lib = _lib.load()
    one = ctypes.c_void_p(16)   # never dereferenced: refused before any launch
    D = ctypes.c_double
    MERGE, SPLIT, COOP = 128, 64, 32

    def call(n, flags, rows=True):
        return lib.sympa_model_train_backward(one, 100, n, one, 2, one, 2, one, 64, None, 0, 0, None, D(1e-5), None, D(1.0), D(1.0),
                                              one, None if rows else one, one if rows else None, None, None, None, None, None, 0,
                                              flags, None)
    assert call(8, MERGE) == -1 and b"MERGE_SRC" in lib.sympa_last_error()
    assert call(7, MERGE) == -1
    assert call(6, MERGE | SPLIT) == -1 and call(5, MERGE | COOP) == -1


def test_radam_entry_points_validate_their_arguments_without_gpu():
    lib = _lib.load()
    one = ctypes.c_void_p(16)   # never dereferenced: validation happens before any launch
    D = ctypes.c_double
    # rows == 0 is a no-op; null state, bad betas, dims outside 1..6 are refused
    assert lib.sympa_radam_step(one, one, one, one, 0, 4, 0, D(0.1), D(0.9), D(0.999), D(1e-7), D(0.0), one, D(1e-5), None, None, None) == 0
    assert lib.sympa_radam_step(one, one, None, one, 8, 4, 0, D(0.1), D(0.9), D(0.999), D(1e-7), D(0.0), one, D(1e-5), None, None, None) == -1
    assert lib.sympa_radam_step(one, one, one, one, 8, 4, 0, D(0.1), D(1.0), D(0.999), D(1e-7), D(0.0), one, D(1e-5), None, None, None) == -1
    assert lib.sympa_radam_step(one, one, one, one, 8, 4, 7, D(0.1), D(0.9), D(0.999), D(1e-7), D(0.0), one, D(1e-5), None, None, None) == -1
    assert lib.sympa_radam_step(one, one, one, one, 8, 7, 0, D(0.1), D(0.9), D(0.999), D(1e-7), D(0.0), one, D(1e-5), None, None, None) == -2
    assert b"dims" in lib.sympa_last_error()
    # the fused step: null Adam state / empty table before anything touches a device
    assert lib.sympa_radam_step_fused(one, one, None, one, one, 8, 4, 0, D(0.1), D(0.9), D(0.999), D(1e-7), D(0.0), D(1e-5), D(1.0), 1,
                                      None, None, None, None, None, None, None, None, 0, one, 64, None, 0, None, None, None, None) == -1
    assert lib.sympa_radam_step_fused(one, one, one, one, one, 0, 4, 0, D(0.1), D(0.9), D(0.999), D(1e-7), D(0.0), D(1e-5), D(1.0), 1,
                                      None, None, None, None, None, None, None, None, 0, one, 64, None, 0, None, None, None, None) == -1


def test_product_path_refuses_cpu_tensors():
    import torch
    from sympa_amd import ops
    z = torch.zeros(2, 2, 3, 3, dtype=torch.float64)
    with pytest.raises(_lib.SympaHipError):
        ops.siegel_dist_forward(z, z)
    with pytest.raises(_lib.SympaHipError):
        ops.model_forward(z, torch.zeros(2, 2, dtype=torch.int64))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsympa_hip.so")
    with pytest.raises(_lib.SympaHipError):
        _lib.load()


def test_integration_md_stub_compiles_and_matches_the_binding():
    """INTEGRATION.md section 3 shows the ctypes stub a maintainer of the reference would paste into
    sympa/manifolds/siegel_manifold.py.  Extract it, execute it (stand-ins for the two names the reference file
    already has in scope) and check its prototypes against this package's own binding, and that it only uses
    attributes the reference's class really has (self.dims, self.metric -- there is no metric_name)."""
    import re
    from abc import ABC
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):text.index("### 3b.")]      # 3b shows the loops around the path, not the ctypes stub
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1
    src = blocks[0]
    assert "metric_name" not in src and "self.metric" in src and "self.dims" in src
    cwd = os.getcwd()
    os.chdir(ROOT)                  # the stub loads the library by its in-tree relative path
    try:
        ns = {"Manifold": type("Manifold", (), {}), "ABC": ABC}   # geoopt's base class stands in
        exec(compile(src, "INTEGRATION.md#3", "exec"), ns)
    finally:
        os.chdir(cwd)
    lib = _lib.load()
    for name in ("sympa_siegel_dist_fwd", "sympa_siegel_dist_bwd"):
        assert list(getattr(ns["_lib"], name).argtypes) == list(getattr(lib, name).argtypes), name
        assert getattr(ns["_lib"], name).restype is getattr(lib, name).restype
    # the class-name -> id table covers exactly the mirror's metric classes, with the C-ABI's ids
    from sympa_amd.manifolds import metrics as mm
    from sympa_amd.ops import METRIC_IDS
    for t in mm.MetricType:
        cls = type(mm.Metric.get(t, 3)).__name__
        assert ns["_METRIC_ID"][cls] == METRIC_IDS[t.value], cls
    assert hasattr(ns["SiegelManifold"], "dist") and ns["SiegelManifold"].model_id == 0


def test_every_translation_unit_passed_the_dpp_hazard_scan():
    """__graft_entry__.build() scans the gfx950 assembly of every .hip unit for the DPP hazards the compiler cannot see in
    inline asm (tools/check_dpp_hazards.py) and rebuilds a unit that fails with the wait states inside the asm statements;
    its report must list every unit as clean in the end."""
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sympa_amd", "csrc")
    report = os.path.join(csrc, "dpp_hazard_report.txt")
    if not os.path.exists(os.path.join(csrc, "libsympa_hip.so")):
        import pytest
        pytest.skip("library not built in this tree")
    # a built tree WITHOUT the report was not built by the one supported path (__graft_entry__.build_hip, which
    # `make -C sympa_amd/csrc` forwards to): it may contain the hazardous units -- a failure, not a skip
    assert os.path.exists(report), "libsympa_hip.so exists but dpp_hazard_report.txt does not: built outside __graft_entry__.build()"
    lines = {l.split()[0]: l.strip() for l in open(report) if l.strip() and not l.startswith("#")}
    units = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    assert sorted(lines) == units
    dpp = {u: l.split(";")[0].strip() for u, l in lines.items()}
    assert all(l.endswith("clean") for l in dpp.values()), [l for l in dpp.values() if not l.endswith("clean")]
    # ... and (round 5, tools/dma_reload_check.py) no LDS-DMA instruction of any unit sits behind a scratch reload: such a reload
    # is followed by s_waitcnt vmcnt(0) and serialises the gather, one round trip per row
    import re
    for u, l in lines.items():
        m = re.search(r"(\d+) LDS-DMA instructions, (\d+) behind a scratch reload", l)
        assert m, f"{u}: no LDS-DMA scan in the report (rebuild with __graft_entry__.build())"
        assert int(m.group(2)) == 0, l
