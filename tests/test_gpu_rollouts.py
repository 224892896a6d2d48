"""Device-resident closed loop (obca_rollouts_* of include/obca_mpc.h) against the per-rollout ``closedLoop``
mirror of the reference's loop (src/closed_loop.py:323-443, pinned by fixtures F1-F8 in tests/test_harness.py);
both use the HIP solver, so equal inputs give equal outputs.  Config C5 of SURVEY.md 8(d)."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-7     # fp64; device cos/sin/atan2 may differ from numpy's in the last bit and the solves amplify that


def _mirror(setting, N, n_steps):
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    cl = closedLoop(setting)
    cl.N_free = cl.N_fix = N
    steps = 0
    while steps < n_steps and not cl.goal_reached():
        steps += 1
        if not cl.step():
            break
    return cl


def _compare(settings, N, n_steps):
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    dr = DeviceRollouts(pack_worlds(copy.deepcopy(settings)), N=N)
    dr.run(n_steps)
    out = {k: v.cpu().numpy() for k, v in dr.read().items()}
    torch.cuda.synchronize()
    for i, st in enumerate(settings):
        cl = _mirror(copy.deepcopy(st), N, n_steps)
        assert out["steps"][i] == cl.k, (i, out["steps"][i], cl.k)
        k = cl.k
        np.testing.assert_allclose(out["x_closed"][i, :k + 1], np.asarray(cl.x_closed)[:k + 1], rtol=0, atol=TOL)
        if k:
            np.testing.assert_allclose(out["u_closed"][i, :k], np.asarray(cl.u_closed), rtol=0, atol=TOL)
            np.testing.assert_allclose(out["T_closed"][i, :k], np.asarray(cl.T_closed), rtol=0, atol=TOL)
            np.testing.assert_allclose(out["x_openloop"][i, :k], np.asarray([x.T for x in cl.x_openLoop]), rtol=0, atol=TOL)
    return out


def test_demo8_on_device_follows_the_mirror():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    out = _compare([problemSetting("demo8")], 6, 8)
    v = out["variant"][0, :out["steps"][0]].tolist()
    assert v[0] == 4 and (6 in v or 8 in v)


def test_monte_carlo_worlds_on_device_follow_the_mirror():
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    out = _compare([make_world_c5(i) for i in range(6)], 5, 8)
    assert out["steps"].sum() >= 30


def test_fused_kernel_equals_lockstep_launches():
    """the persistent one-wave-per-rollout kernel and the per-step launches run the same code on the same data"""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    w = pack_worlds([make_world_c5(i) for i in range(64)])
    outs = []
    for mode in ("fused", "lockstep"):
        dr = DeviceRollouts(w, N=5)
        dr.set_mode(mode)
        dr.run(12)
        outs.append({k: v.cpu().numpy() for k, v in dr.read().items()})
        torch.cuda.synchronize()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert outs[0]["steps"].sum() > 500


def test_closed_loop_batch_properties():
    """C5 at a few hundred rollouts, all 30 steps: every recorded step obeys the unicycle update with the recorded
    input and step length, starts where the previous one ended, respects bounds; bookkeeping is consistent."""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    B = 256
    dr = DeviceRollouts([make_world_c5(i) for i in range(B)], N=5)
    assert dr.queue_mode() == 2                                     # the default on MI355X: one work queue per XCD
    dr.run()
    o = {k: v.cpu().numpy() for k, v in dr.read().items()}
    torch.cuda.synchronize()
    steps, flags = o["steps"], o["flags"]
    assert np.all(flags != 0)                                       # every rollout ended: goal, cap or failure
    assert np.all(steps[flags == 2] == 30)
    assert (flags != 3).mean() > 0.895                              # measured at 4096 rollouts: 92.3 % (316 stopped, 312 of them genuinely infeasible: profiles/r04_bench_classify_all.json)
    done = 0
    for i in range(B):
        k = steps[i]
        x, u, T = o["x_closed"][i], o["u_closed"][i], o["T_closed"][i]
        for j in range(k):
            nxt = x[j] + T[j] * np.array([u[j, 0] * np.cos(x[j, 2]), u[j, 0] * np.sin(x[j, 2]), u[j, 1]])
            assert np.max(np.abs(nxt - x[j + 1])) < 1e-6, (i, j)
            assert np.allclose(o["x_openloop"][i, j, :, 0], x[j], atol=1e-7)
            assert o["variant"][i, j] in (4, 6, 8)
            done += 1
        assert np.all(o["variant"][i, k + 1:] == 0)
        assert np.all(np.abs(u[:k, 0]) <= 0.6 + 1e-6) and np.all(np.abs(u[:k, 1]) <= np.pi / 6 + 1e-6)
        assert np.all((x[:k + 1, 1] > 1.0) & (x[:k + 1, 1] < 9.0))
    assert done > 0.93 * 30 * B
    # a second run from reset reproduces the first bit for bit
    dr.reset()
    dr.run()
    o2 = dr.read()
    torch.cuda.synchronize()
    assert np.array_equal(o2["x_closed"].cpu().numpy(), o["x_closed"])


def test_warm_start_option_on_device():
    """warm start (not reference behaviour, off by default): same closed-loop trajectories on static worlds, fewer
    interior-point iterations; the fused and the lock-step driver agree bit for bit with it as well"""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    w = pack_worlds([make_world_c5(i, n_dyn=0) for i in range(128)])
    cold = {k: v.cpu().numpy() for k, v in DeviceRollouts(w, N=5).run(15).read().items()}
    outs = []
    for mode in ("fused", "lockstep"):
        dr = DeviceRollouts(w, N=5, warm_start=0.1)
        dr.set_mode(mode)
        outs.append({k: v.cpu().numpy() for k, v in dr.run(15).read().items()})
        torch.cuda.synchronize()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    warm = outs[0]
    both = (cold["steps"] == 15) & (warm["steps"] == 15)
    assert both.mean() > 0.9
    assert np.max(np.abs(warm["x_closed"][both, :16] - cold["x_closed"][both, :16])) < 1e-4
    assert warm["iters"][both, 1:15].mean() < 0.9 * cold["iters"][both, 1:15].mean()      # (cold = the window first since round 5: ~16 against ~12 iterations)
    # and with moving obstacles it must still produce valid closed loops
    w2 = pack_worlds([make_world_c5(i, n_dyn=2) for i in range(128)])
    o2 = {k: v.cpu().numpy() for k, v in DeviceRollouts(w2, N=5, warm_start=0.1).run().read().items()}
    torch.cuda.synchronize()
    assert (o2["flags"] != 3).mean() > 0.6 and np.all(o2["flags"] != 0)


def test_long_horizon_rollouts_take_the_lockstep_path():
    """N = 12: more rows than one wavefront holds, so the fused kernel does not apply and obca_rollouts_run falls back to
    per-step launches, whose solves run on the four-wavefront kernel; checked against the CPU build of the same cores"""
    import torch
    from oracle import c_oracle
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    w = pack_worlds([make_world_c5(i, n_dyn=0) for i in range(4)])
    dr = DeviceRollouts(w, N=12)
    dr.run(3)
    out = {k: v.cpu().numpy() for k, v in dr.read().items()}
    torch.cuda.synchronize()
    ref = native_build.rollout_run(w, 12, c_oracle.default_params(), 3)
    assert np.array_equal(out["steps"], ref["steps"]) and out["steps"].sum() >= 8
    assert np.array_equal(out["variant"], ref["variant"])
    np.testing.assert_allclose(out["x_closed"], ref["x_closed"], rtol=0, atol=1e-6)


def test_fixed_time_horizon_twice_the_free_time_one_on_the_device():
    """H5 with resampling ratio 2 (N_free = 5, N_fix = 10) on the GPU: the fused kernel and the lock-step launches equal the
    CPU build of the same harness + structured core, which tests/test_rollout_core.py holds against the Python mirror"""
    from oracle import c_oracle
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    settings = [problemSetting("demo8")] + [make_world_c5(i, n_dyn=1) for i in range(7)]
    ref = None
    for mode in ("fused", "lockstep"):
        groups = {}
        for s in settings:
            groups.setdefault((tuple(int(v) for v in s.static_vObs), len(s.dyn_obs_info)), []).append(s)
        for key, ss in groups.items():
            w = pack_worlds(copy.deepcopy(ss))
            dr = DeviceRollouts(w, N=5, N_fix=10)
            dr.set_mode(mode)
            o = {k: v.cpu().numpy() for k, v in dr.run(6).read().items()}
            h = native_build.rollout_run(pack_worlds(copy.deepcopy(ss)), 5, c_oracle.default_params(), 6, N_fix=10)
            assert o["x_openloop"].shape[3] == 11
            assert np.array_equal(o["steps"], h["steps"]) and np.array_equal(o["variant"][:, :6], h["variant"][:, :6])
            assert (o["variant"] >= 6).any() or len(ss) == 1
            np.testing.assert_allclose(o["x_closed"][:, :7], h["x_closed"][:, :7], rtol=0, atol=1e-6)
            np.testing.assert_allclose(o["x_openloop"][:, :6], h["x_openloop"][:, :6], rtol=0, atol=1e-6)


def test_history_export_on_the_device_matches_the_mirror_lists():
    """row N4 on the GPU: DeviceRollouts.read() -> rollouts.reference_lists gives the lists the reference's closedLoop hands to
    its plot routine (x_openLoop, x_closed, u_closed, Ts_opt, dyn_loc; src/closed_loop.py:416-441, src/draw.py:333-456),
    equal to what the Python mirror of the loop collects with the same solver code on the CPU"""
    import copy
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds, reference_lists
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    settings = [make_world_c5(i) for i in range(4)]
    w = pack_worlds(copy.deepcopy(settings))
    out = DeviceRollouts(w, N=5).run(5).read()
    for i, st in enumerate(settings):
        cl = closedLoop(copy.deepcopy(st), solver=native_build.LpiObca())
        cl.N_free = cl.N_fix = 5
        n = 0
        while n < 5 and not cl.goal_reached():
            n += 1
            if not cl.step():
                break
        lists = reference_lists(out, w, i)
        assert len(lists["x_openLoop"]) == len(cl.x_openLoop) == cl.k
        for a, b in zip(lists["x_openLoop"], cl.x_openLoop):
            assert a.shape == np.asarray(b).shape
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array(lists["x_closed"]), np.array(cl.x_closed), rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array(lists["u_closed"]), np.array(cl.u_closed), rtol=0, atol=1e-6)
        np.testing.assert_allclose(lists["Ts_opt"], cl.T_closed, rtol=0, atol=1e-7)
        assert len(lists["dyn_loc"]) == len(cl.dyn_loc)
        for a, b in zip(lists["dyn_loc"], cl.dyn_loc):
            assert len(a) == len(b)
            for va, vb in zip(a, b):
                np.testing.assert_allclose(np.array(va[:5]), np.array(vb[:5]), rtol=0, atol=1e-9)
                assert va[5] == vb[5]


def test_step_queue_schedule_equals_one_workgroup_per_rollout(monkeypatch):
    """the fused kernel's schedules hand a rollout from workgroup to workgroup after every round: the default
    (OBCA_ROLLOUT_QUEUE=2) within the rollout's XCD, whose L2 all its CUs share -- no write-back, only the L1 invalidate of the
    agent-scope acquire -- and the global queue (=1) from XCD to XCD through HBM with an agent-scope release / acquire; with
    OBCA_ROLLOUT_QUEUE=0 one workgroup keeps the rollout for life.  Same arithmetic on the same data: EVERY word of every
    output must be equal -- 2048 rollouts of uneven cost (two moving boxes), all 30 steps, i.e. ~20 000 hand-offs per schedule
    under load with warm L1s; then once more with the optional warm start, whose primal vectors travel the same way."""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    w = pack_worlds([make_world_c5(i, n_dyn=2) for i in range(2048)])
    for warm in (None, 0.1):
        outs = []
        for env in ("2", "1", "0"):
            monkeypatch.setenv("OBCA_ROLLOUT_QUEUE", env)
            dr = DeviceRollouts(w, N=5, warm_start=warm) if warm else DeviceRollouts(w, N=5)
            assert dr.queue_mode() == int(env)       # MI355X reports eight XCCs: the per-XCD queues are offered (and the default)
            dr.run()
            outs.append({k: v.cpu().numpy() for k, v in dr.read().items()})
            torch.cuda.synchronize()
        # the safety net of the per-XCD schedule: an XCD that received no workgroup leaves its queue to the global queue's
        # clean-up pass.  Placement cannot be forced, so the hook below cuts the per-XCD launch short after 12 steps instead:
        # the clean-up pass then finds the first two rounds (of six steps) done and does the other three itself, across the
        # launch boundary
        monkeypatch.setenv("OBCA_ROLLOUT_QUEUE", "2")
        monkeypatch.setenv("OBCA_ROLLOUT_LOCAL_STEPS", "12")
        dr = DeviceRollouts(w, N=5, warm_start=warm) if warm else DeviceRollouts(w, N=5)
        dr.run()
        outs.append({k: v.cpu().numpy() for k, v in dr.read().items()})
        torch.cuda.synchronize()
        monkeypatch.delenv("OBCA_ROLLOUT_LOCAL_STEPS")
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[2][k]), ("per-XCD queues", warm, k)
            assert np.array_equal(outs[1][k], outs[2][k]), ("global queue", warm, k)
            assert np.array_equal(outs[3][k], outs[2][k]), ("per-XCD queues + clean-up pass", warm, k)
        assert outs[0]["steps"].sum() > 40000


def test_terminal_screen_and_dodge_leave_the_closed_loop_words_or_say_why():
    """include/obca_mpc.h: terminal_screen -- obca_mpc6 calls that cannot reach their terminal set are answered without a solve.  The
    driver discards a failed obca_mpc6 (src/closed_loop.py:393-398), so with the screen on or off every pose, input, step length,
    variant and status of every rollout is the same word; only the iteration counts drop (the screened solves are not run)."""
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    w = pack_worlds([make_world_c5(i) for i in range(256)])
    outs = {}
    for mode in ("fused", "lockstep"):
        for screen in (True, False):
            dr = DeviceRollouts(w, N=5, params=SolverParams(terminal_screen=screen))
            dr.set_mode(mode)
            dr.run(30)
            outs[mode, screen] = {k: v.cpu().numpy() for k, v in dr.read().items()}
            torch.cuda.synchronize()
    ref = outs["fused", True]
    for key, o in outs.items():
        for k in ref:
            if k != "iters":
                assert np.array_equal(ref[k], o[k]), (key, k)
    assert np.array_equal(ref["iters"], outs["lockstep", True]["iters"]) and np.array_equal(outs["fused", False]["iters"], outs["lockstep", False]["iters"])
    it_on, it_off = int(ref["iters"].sum()), int(outs["fused", False]["iters"].sum())
    assert it_on < 0.9 * it_off, (it_on, it_off)


def test_second_dodge_level_keeps_rollouts_alive_on_device_as_on_the_host():
    """csrc/obca_device.h: OBCA_DODGE_LEVEL2_MU.  C5 worlds 667 (named as a solver failure in round 5's bench line, step 20), 7225 and 11188
    (the only solver failures of 8192 held-out worlds, profiles/r06_c5_heldout_failures.json): obca_mpc8 ends feasible through the second
    level of the dodge rung, the rollouts go on -- fused kernel, lock-step launches and the host build of the harness agree on every step;
    with the rung off they stop where round 5 stopped."""
    import torch
    from oracle import c_oracle
    from tests import native_build
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    ids, stop_r5 = [667, 7225, 11188], [20, 16, 13]
    worlds = [make_world_c5(i, n_dyn=2) for i in ids]
    res = []
    for mode in ("fused", "lockstep"):
        dr = DeviceRollouts(pack_worlds(worlds), N=5)
        dr.set_mode(mode)
        dr.run(30)
        torch.cuda.synchronize()
        res.append({k: v.cpu().numpy() for k, v in dr.read().items()})
    for k in ("steps", "flags", "variant", "status", "x_closed", "T_closed"):
        assert np.array_equal(res[0][k], res[1][k]), k
    host = native_build.rollout_run(pack_worlds(worlds), 5, c_oracle.default_params(), 30)
    for j, b in enumerate(ids):
        k = int(res[0]["steps"][j])
        assert k > stop_r5[j] and host["steps"][j] == k, (b, k, host["steps"][j])
        assert res[0]["variant"][j, stop_r5[j]] == 8 and res[0]["status"][j, stop_r5[j]] in (0, 1)
        np.testing.assert_allclose(res[0]["x_closed"][j, :k + 1], host["x_closed"][j, :k + 1], rtol=0, atol=1e-6)
    w = pack_worlds(worlds)
    dr = DeviceRollouts(w, N=5, params=SolverParams(xL=getattr(w, "xL", (0.0, 0.0)), xU=getattr(w, "xU", (39.0, 10.0)), dodge=False))
    dr.run(30)
    torch.cuda.synchronize()
    off = {k: v.cpu().numpy() for k, v in dr.read().items()}
    assert off["steps"].tolist() == stop_r5 and (off["flags"] == 3).all()
