"""The C-ABI library builds, loads and exports every symbol include/obca_mpc.h declares (no GPU calls)."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as ge
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib


@pytest.fixture(scope="module")
def lib():
    ge.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ge.ROOT, "include", "obca_mpc.h")).read()
    declared = set(re.findall(r"\b(obca_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_strerror_and_version(lib):
    assert lib.obca_strerror(0) == b"ok"
    assert b"LDS" in lib.obca_strerror(-28)
    assert b"gfx950" in lib.obca_version()


def test_lds_budget(lib):
    d = _lib.ObcaDims()
    d.N, d.n_obs, d.max_batch, d.device = 5, 3, 8, 0
    d.m[0], d.m[1], d.m[2] = 1, 4, 1
    small = lib.obca_lds_bytes(ctypes.byref(d))
    assert 0 < small <= 64 * 1024
    d.N, d.n_obs = 5, 5
    for i, v in enumerate([2, 4, 4, 2, 2]):
        d.m[i] = v
    assert small < lib.obca_lds_bytes(ctypes.byref(d)) <= 160 * 1024
    d.m[0] = 9                                   # more edges than the compiled limit
    assert lib.obca_lds_bytes(ctypes.byref(d)) == -1


def test_create_rejects_bad_dims(lib):
    d = _lib.ObcaDims()
    d.N, d.n_obs, d.max_batch, d.device = 0, 3, 8, 0
    h = ctypes.c_void_p()
    assert lib.obca_create(ctypes.byref(d), ctypes.byref(h)) == -22
    d.N, d.n_obs = 40, 8
    for i in range(8):
        d.m[i] = 4
    assert lib.obca_lds_bytes(ctypes.byref(d)) > 160 * 1024          # beyond the LDS kernel: lane kernel only


def test_no_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BatchSolver(5, [1, 4, 1], 4)


def test_header_is_plain_c(tmp_path):
    """include/obca_mpc.h is the FFI contract: it must compile as C99 without HIP or C++"""
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "obca_mpc.h"\nint main(void) { obca_dims d; (void)d; return (int)sizeof(obca_params) == 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                    "-I", os.path.join(ge.ROOT, "include"), str(src)], check=True)
