"""Edge cases of the C ABI on the device: empty and masked batches, argument errors, the compiled size limits, the
three-box (M=12) sub-configuration of C2, instance skipping."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu


def _dev(b, keys=("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")):
    return {k: torch.as_tensor(b[k], device="cuda") for k in keys}


def test_empty_batch_and_argument_errors():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    b = sc.make_batch(4, 5)
    s = BatchSolver(5, b["m"], max_batch=4)
    d = _dev(b)
    out = s.solve(d["variant"], d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"])
    torch.cuda.synchronize()
    assert out.status.cpu().tolist() == [0, 0, 0, 0]
    lib, cp = s.lib, SolverParams().to_c()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    args = lambda B, x0: (s._h, p(d["variant"]), B, x0, p(d["u0"]), p(d["xref"]), p(d["A"]), p(d["b"]), p(d["Ts"]), p(d["term"]),
                          ctypes.byref(cp), p(out.xopt), p(out.uopt), p(out.ts_opt), p(out.status), p(out.iters), None, None)
    assert lib.obca_solve_batch(*args(0, p(d["x0"]))) == 0                      # empty batch: nothing to do
    assert lib.obca_solve_batch(*args(5, p(d["x0"]))) == -22                    # beyond max_batch
    assert lib.obca_solve_batch(*args(-1, p(d["x0"]))) == -22
    assert lib.obca_solve_batch(*args(4, None)) == -22                          # NULL input
    # obca_params.struct_size (obca_mpc 0.5): a struct that was never initialised, or one of another layout, is refused instead of
    # being read field by field as something it is not; obca_params_init() leaves what a new ObcaParams() holds
    good = cp.struct_size
    assert good == ctypes.sizeof(_lib.ObcaParams)
    for bad in (0, good - 8, good + 8):
        cp.struct_size = bad
        assert lib.obca_solve_batch(*args(4, p(d["x0"]))) == -22
    cp.struct_size = good
    assert lib.obca_solve_batch(*args(4, p(d["x0"]))) == 0
    raw = _lib.ObcaParams()
    ctypes.memset(ctypes.byref(raw), 0xff, ctypes.sizeof(raw))
    lib.obca_params_init(ctypes.byref(raw))
    assert raw.struct_size == good and bytes(raw)[4:] == bytes(ctypes.sizeof(raw) - 4)
    for field, v in (("dodge", 7), ("terminal_screen", -3)):                   # any positive value: on, any negative: off
        setattr(cp, field, v)
        assert lib.obca_solve_batch(*args(4, p(d["x0"]))) == 0
    with pytest.raises(ValueError):
        s.solve(d["variant"], d["x0"][:, :2], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"])
    assert lib.obca_set_mode(s._h, 7) == -22
    assert lib.obca_set_two_sided_sweep(s._h, 2) == -22 and lib.obca_set_two_sided_sweep(None, 0) == -22
    assert lib.obca_set_two_sided_sweep(s._h, -1) == 0
    assert lib.obca_set_warm_start(s._h, p(out.xopt), None, 0.0) == -22         # mu_init must be positive
    s.close()


def test_masked_instances_are_skipped():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver
    b = sc.make_batch(8, 5)
    for mode in ("wave", "lane"):
        s = BatchSolver(5, b["m"], max_batch=8, mode=mode)
        d = _dev(b)
        full = s.solve(d["variant"], d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"])
        torch.cuda.synchronize()
        ref = full.xopt.clone()
        var = d["variant"].clone()
        var[1::2] = 0
        out = s.solve(var, d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"])
        out.xopt[1::2] = -7.0
        out = s.solve(var, d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"], out=out)
        torch.cuda.synchronize()
        assert out.status.cpu().tolist()[1::2] == [-5] * 4 and out.iters.cpu().tolist()[1::2] == [0] * 4
        assert torch.all(out.xopt[1::2] == -7.0)                                 # outputs untouched
        assert torch.equal(out.xopt[0::2], ref[0::2])                            # neighbours unaffected, bit for bit
        s.close()


def test_three_box_subconfiguration_matches_oracle():
    """C2's second sub-configuration: three 4-row boxes, M = 12"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver
    B = 24
    b = sc.make_batch(B, 5, three_boxes=True)
    assert sum(b["m"]) == 12
    s = BatchSolver(5, b["m"], max_batch=B)
    d = _dev(b)
    out = s.solve(d["variant"], d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"])
    torch.cuda.synchronize()
    ref = c_oracle.solve_batch(4, 5, b["m"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], None, threads=8)
    st = out.status.cpu().numpy()
    assert np.array_equal(st, ref["status"])
    it = out.iters.cpu().numpy()
    for i in range(B):
        if st[i] not in (0, 1):
            continue
        tol = 1e-9 if it[i] == ref["iters"][i] else 1e-5
        assert np.max(np.abs(out.xopt[i].cpu().numpy() - ref["xopt"][i])) < tol
        assert abs(out.ts_opt[i].item() - ref["ts_opt"][i]) < tol
    assert np.mean(it == ref["iters"]) > 0.8
    s.close()


def test_compiled_size_limits():
    """8 obstacles x 4 edges is the largest shape the ABI takes; it runs (on whichever kernel holds it) and every
    instance ends with a defined status; one more obstacle or edge is refused at create."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver
    rng = np.random.default_rng(3)
    N, m, B = 3, [4] * 8, 4
    M = sum(m)
    A = np.zeros((B, N + 1, M, 2)); bb = np.zeros((B, N + 1, M))
    for i in range(8):                                  # small boxes far from the path (x >= 20)
        cx, cy = 20.0 + 2.0 * i, 8.5
        rows = np.array([[1, 0, cx + .3], [-1, 0, -(cx - .3)], [0, 1, cy + .3], [0, -1, -(cy - .3)]], float)
        A[:, :, 4 * i:4 * i + 4] = rows[:, :2]
        bb[:, :, 4 * i:4 * i + 4] = rows[:, 2]
    x0 = np.tile([3.0, 4.0, 0.0], (B, 1)) + rng.uniform(-0.1, 0.1, (B, 3))
    xref = np.zeros((B, 3, N + 1)); xref[:, 0] = 3.0 + np.arange(N + 1); xref[:, 1] = 4.0
    s = BatchSolver(N, m, max_batch=B)
    out = s.solve(4, x0, np.zeros((B, 2)), xref, A, bb, np.full(B, 0.1))
    torch.cuda.synchronize()
    assert set(out.status.cpu().tolist()) <= {0, 1}
    assert np.max(np.abs(out.xopt[:, :, -1].cpu().numpy() - xref[:, :, -1])) < 1e-6      # terminal equality
    s.close()
    d = _lib.ObcaDims()
    d.N, d.n_obs, d.max_batch, d.device = 3, 9, 4, 0
    h = ctypes.c_void_p()
    assert s.lib.obca_create(ctypes.byref(d), ctypes.byref(h)) == -22
    d.n_obs = 1
    d.m[0] = 5
    assert s.lib.obca_create(ctypes.byref(d), ctypes.byref(h)) == -22


def test_c4_total_size_on_one_gpu():
    """C4 is 65 536 instances over 8 GPUs; the whole of it also fits one (86 MB of inputs and outputs).  65 536 UNIQUE seeds
    (generator in parallel processes): every instance converges, its plan ends on the window's last pose, and the eight
    contiguous shards an 8-GPU run would solve (sharding.shard_bounds) give, solved one after the other on this GPU, the
    same words as the one big launch -- an instance's answer does not depend on where in which batch it sits."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.sharding import shard_bounds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver
    B, world = 65536, 8
    b = sc.make_batch(B, 5, procs=max(1, min(64, os.cpu_count() or 1)))
    assert len(np.unique(b["x0"], axis=0)) == B
    s = BatchSolver(5, b["m"], max_batch=B)
    d = _dev(b)
    out = s.solve(d["variant"], d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"])
    torch.cuda.synchronize()
    ok = (out.status == 0) | (out.status == 1)
    assert int(ok.sum()) >= B - 8, int(ok.sum())                      # measured: 65 536 of 65 536
    end = (out.xopt[:, :, -1] - d["xref"][:, :, -1]).abs().amax(dim=1)
    assert float(end[ok].max()) < 1e-6                               # terminal equality
    xo, st, it = out.xopt.clone(), out.status.clone(), out.iters.clone()
    s.close()
    s8 = BatchSolver(5, b["m"], max_batch=B // world)
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        o = s8.solve(*[d[k][lo:hi].contiguous() for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")])
        torch.cuda.synchronize()
        assert torch.equal(o.xopt, xo[lo:hi]) and torch.equal(o.status, st[lo:hi]) and torch.equal(o.iters, it[lo:hi]), r
    s8.close()


def test_status_and_iterations_do_not_depend_on_the_kernel():
    """the same NLP must not end with a different status because auto mode / OBCA_MODE / set_mode picked another
    kernel.  The filter capacity is a function of the problem shape (csrc/obca_device.h: OBCA_FILTER_CAP), so the three
    kernels stop at the same point when it fills up.  2048 C2 instances (which hold line-search and filter-overflow
    cases) through modes 1 / 3 / 2: identical status everywhere, identical iteration counts on all but a handful of
    roundoff-sensitive paths."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    B, N = 2048, 5
    b = sc.make_batch(B, N)
    res = {}
    for mode in ("wave", "multiwave", "lane"):
        s = BatchSolver(N, b["m"], max_batch=B, mode=mode)
        o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
        torch.cuda.synchronize()
        res[mode] = (o.status.cpu().numpy(), o.iters.cpu().numpy())
        s.close()
    for mode in ("multiwave", "lane"):
        feas_a, feas_b = np.isin(res["wave"][0], (0, 1)), np.isin(res[mode][0], (0, 1))
        assert np.array_equal(feas_a, feas_b), (mode, np.flatnonzero(feas_a != feas_b))
        assert (res["wave"][0] != res[mode][0]).sum() <= 2              # failure KIND may differ on a roundoff-sensitive path
        assert (res["wave"][1] != res[mode][1]).mean() < 0.02


def test_invalid_variants_are_per_instance_errors():
    """variant outside {0, 4, 6, 8} and obca_mpc6 without a terminal set: status OBCA_STATUS_BAD_VARIANT (-6), outputs
    untouched, the neighbours solved (the C ABI used to solve them as something else / dereference NULL)"""
    import ctypes
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    b = sc.make_batch(4, 5)
    for mode in ("wave", "lane"):
        s = BatchSolver(5, b["m"], max_batch=4, mode=mode)
        var = np.array([4, 7, 4, -1], np.int32)
        o = s.solve(var, b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
        torch.cuda.synchronize()
        assert o.status.cpu().tolist() == [0, _lib.STATUS_BAD_VARIANT, 0, _lib.STATUS_BAD_VARIANT]
        # obca_mpc6 with term == NULL through the raw C ABI
        dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda")
        t = [dev(np.array([6, 4, 0, 4], np.int32), torch.int32)] + [dev(b[k], torch.float64) for k in ("x0", "u0", "xref", "A", "b", "Ts")]
        outs = [torch.zeros(4, 3, 6, dtype=torch.float64, device="cuda"), torch.zeros(4, 2, 5, dtype=torch.float64, device="cuda"),
                torch.zeros(4, dtype=torch.float64, device="cuda"), torch.zeros(4, dtype=torch.int32, device="cuda"),
                torch.zeros(4, dtype=torch.int32, device="cuda")]
        p = SolverParams().to_c()
        ptr = lambda x: ctypes.c_void_p(x.data_ptr())
        rc = s.lib.obca_solve_batch(s._h, ptr(t[0]), 4, *[ptr(x) for x in t[1:]], None, ctypes.byref(p), *[ptr(x) for x in outs],
                                    None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert rc == 0 and outs[3].cpu().tolist() == [_lib.STATUS_BAD_VARIANT, 0, _lib.STATUS_SKIPPED, 0]
        assert float(outs[0][0].abs().max()) == 0.0
        s.close()


def test_hbm_workspace_kernel_is_bit_identical_to_the_lds_kernel():
    """mode "global" (obca_ipm_kernel_gm: four wavefronts per instance, row state and every O(rows) array in an HBM workspace,
    only the O(N) blocks of the Riccati sweep in LDS -- the kernel auto mode picks beyond the LDS) on shapes the LDS-resident
    four-wavefront kernel also runs: same code, same thread <-> row assignment, same arithmetic => identical output words"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    # gated at N = 16 / 19 / 20: 899 / 1058 / 1114 rows on obca_ipm_kernel_mw_r5, whose fifth row slot (rows >= 1024) lives in
    # LDS -- empty at N = 16, 34 and 90 rows at N = 19 and 20
    for b, N in ((sc.make_batch(128, 5), 5), (sc.make_batch_c3(48, 20, gated=False), 20), (sc.make_batch_c3(48, 20, gated=True), 20),
                 (sc.make_batch_c3(24, 16, gated=True), 16), (sc.make_batch_c3(24, 19, gated=True), 19)):
        res = []
        for mode in ("multiwave", "global"):
            s = BatchSolver(N, b["m"], max_batch=len(b["variant"]), mode=mode)
            o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
            torch.cuda.synchronize()
            res.append((o.xopt.cpu().numpy(), o.uopt.cpu().numpy(), o.ts_opt.cpu().numpy(), o.status.cpu().numpy(), o.iters.cpu().numpy()))
            s.close()
        for a, c in zip(*res):
            assert np.array_equal(a, c)


def test_auto_mode_picks_the_hbm_workspace_kernel_beyond_the_lds():
    """N = 26 with five obstacles (1429 rows, 178 KB of LDS if it were resident: beyond both limits of the LDS-resident
    kernels): a batch through auto mode and through the lane kernel -- same verdicts, same plans where the iteration counts
    agree.  (The C3 generator needs seconds per instance beyond N = 26 -- its lattice paths are ~40 points long -- so the
    horizon stays there; N = 40 ... 74 run in tests/test_gpu_open_loop.py.)"""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    N, B = 26, 32
    b = sc.make_batch_c3(B, N, gated=True)
    res = {}
    for mode in (None, "lane"):
        s = BatchSolver(N, b["m"], max_batch=B, mode=mode)
        if mode is None:
            assert s.lds_bytes > 160 * 1024
        o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
        torch.cuda.synchronize()
        res[mode] = (o.xopt.cpu().numpy(), o.status.cpu().numpy(), o.iters.cpu().numpy())
        s.close()
    ok_a, ok_l = np.isin(res[None][1], (0, 1)), np.isin(res["lane"][1], (0, 1))
    assert ok_a.mean() > 0.9 and (ok_a != ok_l).sum() <= 2
    same = ok_a & ok_l & (res[None][2] == res["lane"][2])
    assert same.sum() >= 0.3 * B and np.abs(res[None][0] - res["lane"][0])[same].max() < 1e-7


def test_workspace_allocation_failure_falls_back_to_the_lds_kernel(monkeypatch):
    """csrc/obca_capi.hip (advisor, round 5): a shape auto mode sends to the one-wavefront HBM-workspace kernel (C3 free-time, N = 12, three
    obstacles) on a handle whose workspace cannot be allocated: auto mode runs the LDS-resident four-wavefront kernel instead of returning
    OBCA_E_NOMEM -- the same words as that kernel asked for by name -- and keeps doing so; an explicit workspace mode reports the failure."""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    B, N = 16, 12
    b = sc.make_batch_c3(B, N, gated=False)
    args = (b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
    ref = BatchSolver(N, b["m"], max_batch=B, mode="multiwave")
    want = ref.solve(*args)
    torch.cuda.synchronize()
    monkeypatch.setenv("OBCA_FAIL_WORKSPACE_ALLOC", "1")
    s = BatchSolver(N, b["m"], max_batch=B)
    for _ in range(2):
        out = s.solve(*args)
        torch.cuda.synchronize()
        for k in ("xopt", "uopt", "ts_opt", "status", "iters"):
            assert torch.equal(getattr(out, k), getattr(want, k)), k
    s.close()
    s = BatchSolver(N, b["m"], max_batch=B, mode="global1")
    with pytest.raises(RuntimeError):
        s.solve(*args)
    s.close()
    monkeypatch.delenv("OBCA_FAIL_WORKSPACE_ALLOC")
    s = BatchSolver(N, b["m"], max_batch=B)                 # without the failure: the workspace kernel, same optimum
    out = s.solve(*args)
    torch.cuda.synchronize()
    assert torch.equal(out.status, want.status) and torch.allclose(out.xopt, want.xopt, rtol=0, atol=1e-6)
    s.close(); ref.close()
