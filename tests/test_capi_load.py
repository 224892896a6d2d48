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


def test_ctypes_mirror_of_obca_params_matches_the_header(lib, tmp_path):
    """the Python mirror (_lib.ObcaParams) against the C header, field by field: a C program prints sizeof and every offsetof; and
    obca_params_init (no GPU needed) leaves all zero + struct_size -- what a new ObcaParams() holds"""
    import subprocess
    fields = [f[0] for f in _lib.ObcaParams._fields_]
    src = tmp_path / "o.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "obca_mpc.h"\nint main(void) {\n  printf("%zu\\n", sizeof(obca_params));\n'
                   + "".join('  printf("%%zu\\n", offsetof(obca_params, %s));\n' % f for f in fields) + "  return 0;\n}\n")
    exe = tmp_path / "o"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ge.ROOT, "include"), str(src), "-o", str(exe)], check=True)
    nums = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(_lib.ObcaParams)
    assert nums[1:] == [getattr(_lib.ObcaParams, f).offset for f in fields]
    assert fields[0] == "struct_size" and fields[-2:] == ["dodge", "terminal_screen"]
    p = _lib.ObcaParams()
    ctypes.memset(ctypes.byref(p), 0x5a, ctypes.sizeof(p))
    lib.obca_params_init(ctypes.byref(p))
    assert p.struct_size == nums[0] and bytes(p)[4:] == bytes(nums[0] - 4)
    assert bytes(_lib.ObcaParams()) == bytes(p)


def test_committed_counter_summary_matches_the_kernel_sources():
    """bench.py reports roofline.traffic / valu_issue_frac only from a profiles/r<NN>_pmc_summary.json whose kernel-source hash is the
    tree's (tools/pmc_summary.py writes the hash of the sources the rocprofv3 --pmc passes measured): the newest matching file is
    picked, and the committed tree has one -- a kernel change without a fresh `tools/profile.sh` run shows up here, not as a silent
    `traffic: null` in the driver's bench line"""
    import os
    import bench
    f = bench._pmc_summary_file()
    assert os.path.exists(f)
    r = bench.pmc_summary("obca_ipm_kernel_s5_3_6", 8192, 5, 6)
    assert r is not None and not r["stale"], (f, r and r.get("source_hash"), bench.kernel_source_hash())
    assert 10.8e6 < r["traffic_bytes"] < 1.4 * 10.81e6                 # 1320 B x 8192 algorithmic; measured 13.2 MB
