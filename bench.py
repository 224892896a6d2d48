#!/usr/bin/env python3
"""bench.py -- OBCA closed-loop MPC steps/sec (batch) at N=5, 3 obstacles on MI355X.

One "step" = one pass of the hot path (one obca_mpc4 solve per instance) over a batch of B=8192 seeded
synthetic instances (SURVEY.md section 8(d), config C2 at the batch size BASELINE.json quotes).  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line (see the repo contract).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Multi-GPU: the batch of independent instances is sharded, one process per GPU, B per GPU fixed (weak scaling);
no collective on the data path, one RCCL all_gather of the outputs after the timed region plus the reductions
needed for timing/reporting.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {(5, 6): 1320, (5, 12): 2184, (6, 6): 1528}       # algorithmic bytes per instance-step
HBM_PEAK_GBPS = 8000.0                                          # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec)




def _pmc_summary_file():
    """the counter summary to report from: the newest profiles/r<NN>_pmc_summary.json taken on the kernel sources in the tree; if none
    matches, the newest one (reported as stale, i.e. not at all)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")), reverse=True)
    for f in files:
        try:
            with open(f) as fh:
                if json.load(fh).get("kernel_source_hash") == kernel_source_hash():
                    return f
        except Exception:          # noqa: BLE001
            continue
    return files[0] if files else os.path.join(ROOT, "profiles", "none")


VALU_LANE_RATE = 256 * 4 * 16 * 2.4e9        # lanes the VALUs of the chip issue per second: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3e12


def kernel_source_hash():
    """hash of the HIP sources the solver kernels are compiled from: the counter passes under profiles/ carry the hash of the
    sources they measured, so that figures of an OLDER kernel are not reported as current (tools/pmc_summary.py writes it)"""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd", "csrc")
    for name in ("obca_kernel.hip", "obca_device.h", "obca_rollout_core.h"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def pmc_summary(kernel, B, N, M):
    """HBM traffic and issue counters of the dominant kernel, from the rocprofv3 --pmc passes committed under profiles/
    (tools/pmc_summary.py turns the counter CSVs into this file).  None when no pass matches this workload; a pass taken on
    other kernel sources than the ones in the tree is returned with `stale` set and its figures are not reported."""
    try:
        with open(_pmc_summary_file()) as f:
            doc = json.load(f)
    except Exception:
        return None
    for r in doc["kernels"]:
        if r.get("kernel") == kernel and (r.get("B"), r.get("N"), r.get("M")) == (B, N, M):
            r = dict(r, source_hash=doc.get("kernel_source_hash"), git_sha=doc.get("git_sha"))
            r["stale"] = r["source_hash"] != kernel_source_hash()
            return r
    return None


def ipopt_leg(batch, N, seconds=10.0):
    """the reference's own solver (CasADi/IPOPT) on the same instances, one core, when casadi imports."""
    from oracle import casadi_ipopt
    if not casadi_ipopt.available():
        return "unavailable"
    from oracle.obca_nlp import Problem
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    n, ok, t0 = 0, 0, time.time()
    while time.time() - t0 < seconds and n < batch["x0"].shape[0]:
        p = Problem(4, N, batch["m"], batch["x0"][n], batch["u0"][n], batch["xref"][n], batch["A"][n], batch["b"][n],
                    sc.TS, 0.1 * np.eye(3), 0.01 * np.eye(2), 0.1 * np.eye(2), 0.1 * np.eye(3), sc.XL, sc.XU,
                    [-0.6, -np.pi / 6], [0.6, np.pi / 6], sc.EGO, sc.DMIN)
        ok += bool(casadi_ipopt.solve(p)["feas"])
        n += 1
    dt = time.time() - t0
    return {"value": n / dt, "unit": "MPC steps/s", "cores": 1, "solved": ok, "sample": "%d instances, CasADi Opti + IPOPT, %.1f s" % (n, dt)}


def alg_bytes(N, M):
    return 8 * (3 + 2 + 3 * (N + 1) + 3 * M * (N + 1) + 1 + 3) + 8 * (3 * (N + 1) + 2 * N + 1) + 8


def cpu_oracle(batch, N, seconds=10.0):
    """The dense oracle (plain-C restatement, oracle/obca_oracle.c: dense Bunch-Kaufman KKT solves) timed on the host cores of
    this box on a bounded sample of the same workload -- context: it is the CHECKER's speed, not a same-algorithm baseline."""
    from oracle import c_oracle
    cores = os.cpu_count() or 1
    n, solved = 0, 0
    chunk = max(cores, 8)
    t0 = time.time()
    while time.time() - t0 < seconds and n + chunk <= batch["x0"].shape[0]:
        sl = slice(n, n + chunk)
        st = c_oracle.solve_batch(4, N, batch["m"], batch["x0"][sl], batch["u0"][sl], batch["xref"][sl],
                                  batch["A"][sl], batch["b"][sl], batch["Ts"][sl], None, threads=cores)["status"]
        solved += int(np.sum((st == 0) | (st == 1)))
        n += chunk
    dt = time.time() - t0
    return {"value": n / dt, "unit": "MPC steps/s", "cores": cores, "solved": solved,
            "sample": "%d instances of the same batch, dense C oracle (oracle/obca_oracle.c), %d threads, %.1f s" % (n, cores, dt)}


def independent_leg(solver, batch, dv, N, prm, n=64):
    """An independent solver (SciPy SLSQP on the reference-pinned model, tests/independent.py) on the first n answers of the
    batch: (a) started at the GPU's answer it must not find a better feasible point; (b) started from the reference window
    and two perturbations of it -- share of starts that end at the same optimum / another / a better one / nowhere."""
    from tests import independent as ind, kkt_check
    solver.enable_certificates()
    out = solver.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], prm)
    torch.cuda.synchronize()
    st = out.status[:n].cpu().numpy()
    z, y = solver.cert_z[:n].cpu().numpy(), solver.cert_y[:n].cpu().numpy()
    ok = [i for i in range(n) if st[i] in (0, 1)]
    ps = {i: kkt_check.problem_of(batch, i, N) for i in ok}
    procs = max(1, min(128, os.cpu_count() or 1))
    t0 = time.time()
    pol = ind.pool_map(ind.polish, [(ps[i], z[i], y[i]) for i in ok], procs)
    sta = ind.pool_map(ind.from_starts, [(ps[i], z[i], i) for i in ok], procs)
    res = ind.summarise(pol, sta)
    res["sample"] = "first %d instances of the batch, SciPy SLSQP on oracle/obca_nlp.py, %d processes, %.1f s" % (n, procs, time.time() - t0)
    return res


def cpu_baseline(batch, N):
    """The CPU baseline: the same algorithm as the GPU kernels -- the structured core csrc/obca_lpi_core.h (the header the lane
    kernel compiles), built for the host by tests/native and run with OpenMP over instances on ALL host cores.  (The
    reference's own solver, CasADi/IPOPT, does not exist in this image: see `ipopt`.)"""
    try:
        from tests import native_build
        native_build.load()
    except Exception as e:          # no g++ / build failed: the figure is optional
        return {"value": None, "note": "host build of the structured core unavailable: %r" % (e,)}
    cores = os.cpu_count() or 1
    n = min(batch["x0"].shape[0], max(64 * cores, 1024))           # one call: the thread pool is started once
    sl = slice(0, n)
    native_build.lpi_solve(4, N, batch["m"], batch["x0"][:cores], batch["u0"][:cores], batch["xref"][:cores],
                           batch["A"][:cores], batch["b"][:cores], batch["Ts"][:cores])          # warm the pool
    t0 = time.time()
    st = native_build.lpi_solve(4, N, batch["m"], batch["x0"][sl], batch["u0"][sl], batch["xref"][sl], batch["A"][sl],
                                batch["b"][sl], batch["Ts"][sl])["status"]
    solved = int(np.sum((st == 0) | (st == 1)))
    dt = time.time() - t0
    return {"value": n / dt, "unit": "MPC steps/s", "cores": cores, "kind": "port", "solved": solved,
            "sample": "%d instances of the same batch, structured core (csrc/obca_lpi_core.h) on the host, %d OpenMP threads, %.1f s" % (n, cores, dt)}


_C3_BATCHES = {}          # generated once per run (the start-order legs solve the same instances)


def config_c3(B, N=20, start_order=0, classify=False):
    """Config C3: N=20, walls + box + two moving boxes, lidar-gated: the free-time sub-batch (obca_mpc4, three
    static obstacles) and the gated sub-batch (obca_mpc6, five obstacles, time-varying rows), B UNIQUE seeded instances
    each; both run on the four-wavefront LDS kernel."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    res = {"workload": "C3 (SURVEY 8d): N=%d, B=%d unique seeded instances per sub-batch" % (N, B),
           "kernel": "free-time half (three obstacles): one wavefront per instance, row state in an HBM workspace (obca_ipm_kernel_gm1); gated half (five obstacles): four wavefronts per instance, LDS resident, two-sided Riccati sweep"}
    procs = max(1, min(48, (os.cpu_count() or 1) // 2))
    for name, gated in (("free_time_obca_mpc4", False), ("gated_obca_mpc6", True)):
        if (B, N, gated) not in _C3_BATCHES:
            _C3_BATCHES[(B, N, gated)] = sc.make_batch_c3(B, N, gated=gated, procs=procs)
        b = _C3_BATCHES[(B, N, gated)]
        s = BatchSolver(N, b["m"], B)
        dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
        out = None
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams(start_order=start_order), out=out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        ok = int(((out.status == 0) | (out.status == 1)).sum())
        st = out.status.cpu().numpy()
        res[name] = {"value": ok / best, "unit": "converged solves/s", "ms_per_launch": best * 1e3, "success_rate": ok / B,
                     "status_counts": {str(k): int((st == k).sum()) for k in np.unique(st)},
                     "mean_ipm_iters": float(out.iters.float().mean()), "lds_bytes": s.lds_bytes}
        s.close()
        bad = np.flatnonzero(~np.isin(st, (0, 1)))
        if classify and len(bad) and len(bad) <= 64:
            # every instance the ladder gave up on: SciPy SLSQP on the pinned model from seven starts (tests/independent.py:classify)
            try:
                from tests import independent as ind, kkt_check
                probs = [kkt_check.problem_of(b, int(i), N) for i in bad]
                rows = ind.pool_map(ind.classify, [(q, ind.trajectory_start(q, q.xref), int(i)) for q, i in zip(probs, bad)], min(len(bad), os.cpu_count() or 1))
                res[name]["failures_classified"] = {"instances": [int(i) for i in bad],
                                                    "feasible_point_exists_solver_failure": [int(r["tag"]) for r in rows if r["feasible_point_found"]],
                                                    "no_feasible_point_found": int(sum(not r["feasible_point_found"] for r in rows)),
                                                    "method": "SciPy SLSQP on the pinned model, seven starts each (window, straight line, window moved 1.5 / 3 m to either side)"}
            except Exception as e:          # noqa: BLE001
                res[name]["failures_classified"] = {"error": repr(e)}
    return res


def open_loop(cases=(("demo9", 10), ("demo9", 74), ("demo1", 10), ("demo1", 74), ("demo9", 66)), start_order="default", classify=False):
    """Row N3: the reference's open-loop free-time plan (closedLoop.mpc_openLoop_freeTime, src/closed_loop.py:113-120) as ONE
    instance through the drop-in `obca` class -- the only timing the reference publishes (src/simulation.py:210-231 calc_time,
    called for demo9 in main.py:28: N = 74 -- the length of demo9's A* route -- 136.69 s, N = 10 3.69 s, hardware unspecified; demo9
    at N = 10 has no feasible point: its time-scale bound allows too short a path).  Second call timed (the first allocates the
    handle's workspace)."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    pub = {("demo9", 10): 3.69, ("demo9", 74): 136.69}
    res = {"workload": "open-loop free-time plan (obca_mpc4, default start ladder, start/goal-only reference), batch of ONE, host call to host result",
           "reference_published_s": {"demo9 N=10": 3.69, "demo9 N=74": 136.69, "source": "src/simulation.py:228-231 (calc_time, main.py:28: demo9), hardware unspecified"}}
    for demo, N in cases:
        s = obca()
        s.start_order = start_order
        cl = closedLoop(problemSetting(demo), solver=s)
        cl.N_free = N
        cl.mpc_openLoop_freeTime()
        torch.cuda.synchronize()
        t = time.perf_counter()
        cl.mpc_openLoop_freeTime()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        res["%s_N%d" % (demo, N)] = {"seconds": dt, "feas": bool(cl.feas), "ipm_iters": s.last["iters"], "status": s.last["status"],
                                     "Ts_opt": float(cl.Ts_opt), "reference_published_s": pub.get((demo, N))}
        if not cl.feas and classify:
            # reported infeasible: the same problem (captured from the host build of the core) to SciPy SLSQP, seven starts
            try:
                from oracle.obca_nlp import Problem
                from tests import independent as ind, native_build
                from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
                cpu = native_build.LpiObca()
                c2 = closedLoop(problemSetting(demo), solver=cpu)
                c2.N_free = N
                c2.mpc_openLoop_freeTime()
                c = cpu.calls[-1]
                sp = SolverParams(xL=c2.xL[:2], xU=c2.xU[:2])
                q = Problem(4, N, c["m"], c["x0"], c["u0"], c["xref"], c["A"], c["b"], c["Ts"], sp.Q_free, sp.R_free[0], sp.R_free[1], sp.P_free,
                            sp.xL, sp.xU, sp.uL, sp.uU, sp.ego, sp.dmin)
                r = ind.classify((q, None, "%s_N%d" % (demo, N), ("window", "line", "right 3", "left 3")))
                res["%s_N%d" % (demo, N)]["classified"] = {"feasible_point_found": r["feasible_point_found"], "smallest_violation": r["viol"], "from_start": r["start"],
                                                          "max_Topt": float(q.Tmax), "host_core_status": c["status"],
                                                          "method": "SciPy SLSQP on the pinned model, four starts: window, straight line, window moved 3 m to either side (tests/independent.py:classify)"}
            except Exception as e:          # noqa: BLE001
                res["%s_N%d" % (demo, N)]["classified"] = {"error": repr(e)}
    # the plan the reference repository shows in images/aStar_vs_openLoopOBCA.png (demo9, N = 50, Q = 0.5 I; fixture
    # tests/golden/reference_openloop_demo9.json): how far are the picture's dots from this build's poses?
    try:
        from tests import reference_openloop
        s = obca()
        s.start_order = start_order
        reference_openloop.plan(s)
        torch.cuda.synchronize()
        t = time.perf_counter()
        cl = reference_openloop.plan(s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        d, _ = reference_openloop.marker_distances(reference_openloop.fixture(), cl.xOpt)
        res["demo9_N50_reference_figure"] = {"seconds": dt, "feas": bool(cl.feas), "ipm_iters": s.last["iters"], "status": s.last["status"],
                                             "Ts_opt": float(cl.Ts_opt), "markers": int(len(d)), "marker_to_pose_max_m": float(d.max()),
                                             "marker_to_pose_mean_m": float(d.mean()), "pixel_m": 0.0911}
    except Exception as e:          # noqa: BLE001
        res["demo9_N50_reference_figure"] = {"error": repr(e)}
    return res


def report_figures_leg():
    """closed-loop frames of the reference's project report (tests/golden/reference_report_figures.json: titles = sum(Ts_opt[:k])
    of runs IPOPT solved) replayed through the product path: how many titles does this build's run show?  Figure 12 = demo1,
    Figure 11 = demo11, both as checked in; the demo11 run is also recorded in images/OBCA_dynObs_demo11.gif, whose closed-loop
    markers tests/golden/reference_gif_demo11.json holds."""
    from tests import reference_report
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    fx = reference_report.fixture()
    out = {"fixture": "tests/golden/reference_report_figures.json, tests/golden/reference_gif_demo11.json", "tolerance_s": reference_report.TIME_TOL}
    for name, setting, n in (("figure12_demo1", reference_report.demo1_setting(), 31), ("figure11_demo11", reference_report.demo11_setting(), 61)):
        t0 = time.perf_counter()
        cum, cl = reference_report.replay(setting, obca(), n)
        titles = sorted(f["spend_time"] for f in fx[name]["frames"])
        hits = reference_report.match(cum, titles)
        out[name] = {"titles": titles, "nearest_step": [k for k, _ in hits], "distance_s": [round(e, 4) for _, e in hits],
                     "titles_shown": int(sum(e <= reference_report.TIME_TOL for _, e in hits)), "steps_run": int(len(cum)), "seconds": time.perf_counter() - t0}
        if name == "figure11_demo11":
            g = reference_report.gif_demo11()
            M, first = np.array(g["markers_xy"]), g["first_marker_is_pose"]
            X = np.asarray(cl.x_closed)[first:first + len(M), :2]
            d = np.hypot(*(X[:len(M)] - M[:len(X)]).T)
            out[name].update({"gif_markers": int(len(M)), "marker_to_pose_max_m": float(d.max()), "marker_to_pose_mean_m": float(d.mean()),
                              "marker_accuracy_m": 0.15, "title_reading_precision_s": reference_report.TIME_TOL, "fourth_title_measured_off_s": reference_report.DEMO11_FOURTH_MEASURED})
    return out


def reference_gif_leg():
    """The one solver output the reference repository holds (its GIF of the demo9 closed loop: sum(Ts_opt[:k]) of 83 chained
    IPOPT solves, fixture tests/golden/reference_gif_demo9.json) replayed through the product path (closedLoop mirror on the
    drop-in obca class, one GPU solve per step): how many consecutive steps show the reference's digits."""
    from tests import reference_gif
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
    ref = np.asarray(reference_gif.fixture()["spend_time"][1:])
    out = {"fixture": "tests/golden/reference_gif_demo9.json (83 steps, titles rounded to 0.01 s)", "tolerance_s": reference_gif.TIME_TOL}
    for name in ("default", "x0", "window", "zeros"):          # include/obca_mpc.h: start_order
        s = obca()
        s.start_order = name
        iters = []
        run = s._run
        def counted(*a, **k):
            r = run(*a, **k)
            iters.append(s.last["iters"])
            return r
        s._run = counted
        t0 = time.perf_counter()
        cum, _, cl = reference_gif.replay(s, 83)
        n = min(len(cum), 83)
        bad = np.where(np.abs(cum[:n] - ref[:n]) > reference_gif.TIME_TOL)[0]
        out["start_order_" + name] = {"steps_run": int(len(cum)), "consecutive_steps_matching_the_reference": int(bad[0]) if len(bad) else n,
                     "steps_matching_in_total": int(n - len(bad)), "mean_ipm_iters": float(np.mean(iters)), "seconds": time.perf_counter() - t0}
    return out


def start_order_leg(solver, dv, out0, B, order, steps=3):
    """another obca_params.start_order on the headline batch ("x0": x0 first, obca_mpc4's default until obca_mpc 0.4; "zeros": the
    reference's literal all-zero start first, the default until obca_mpc 0.1): throughput, and how many instances end at the
    optimum the default order finds."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
    prm1 = SolverParams(start_order=order)
    x0, ts0, st0 = out0.xopt.clone(), out0.ts_opt.clone(), out0.status.clone()
    go = lambda: solver.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], prm1)
    o = go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        o = go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = (o.status == 0) | (o.status == 1)
    ok0 = (st0 == 0) | (st0 == 1)
    both = ok & ok0
    same = both & ((o.ts_opt - ts0).abs() <= 1e-6 * ts0.abs().clamp(min=1.0)) & ((o.xopt - x0).abs().amax(dim=(1, 2)) <= 1e-5)
    return {"value": float(ok.sum().item() / dt), "unit": "MPC steps/s", "ms_per_launch": dt * 1e3, "success_rate": float(ok.float().mean().item()),
            "mean_ipm_iters": float(o.iters.float().mean().item()),
            "same_optimum_as_the_default_order": int(same.sum().item()), "of_instances_both_solved": int(both.sum().item())}


def closed_loop_c5(B, n_dyn=2, warm_start=None, first=0, dist=None, classify=False, classify_max=None, start_order=0, dodge=True):
    """Config C5: B Monte-Carlo rollouts of the receding-horizon loop per GPU, harness and solves on the device
    (obca_rollouts_run: one persistent kernel, one wavefront per rollout); worlds first .. first+B-1, resident in HBM
    before the clock starts.  With a process group every rank runs the whole loop for its own worlds (no collective on
    the data path); the time is the max over ranks, the steps are summed."""
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    w = pack_worlds([sc.make_world_c5(first + i, n_dyn=n_dyn) for i in range(B)])
    prm = None
    if start_order or not dodge:
        from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
        prm = SolverParams(xL=getattr(w, "xL", (0.0, 0.0)), xU=getattr(w, "xU", (39.0, 10.0)), start_order=start_order, dodge=dodge)
    dr = DeviceRollouts(w, N=5, warm_start=warm_start, params=prm)
    dr.run(1)
    torch.cuda.synchronize()
    dr.reset()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dr.run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    o = {k: v.cpu().numpy() for k, v in dr.read().items()}
    ok, tried = int(o["steps"].sum()), int((o["variant"] > 0).sum())
    world = 1
    if dist:
        world = dist.get_world_size()
        acc = torch.tensor([float(ok), float(tried), float((o["flags"] == 2).sum()), float((o["flags"] == 3).sum())], device="cuda", dtype=torch.float64)
        tm = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ok, tried, dt = int(acc[0]), int(acc[1]), float(tm[0])
        caps, fails = int(acc[2]), int(acc[3])
    else:
        caps, fails = int((o["flags"] == 2).sum()), int((o["flags"] == 3).sum())
    res = {"workload": "C5 (SURVEY 8d): %d closed-loop rollouts%s, N=5, walls + random box + %d moving 3x3 boxes (lidar gate 10 m), "
                       "<=30 steps each, obca_mpc4 / obca_mpc6 -> obca_mpc8 as the reference dispatches them; harness and "
                       "solves in one persistent kernel (obca_rollouts_run)" % (B * world, " (%d per GPU)" % B if world > 1 else "", n_dyn),
           "value": ok / dt, "unit": "converged closed-loop MPC steps/s", "seconds": dt, "converged_steps": ok,
           "attempted_steps": tried, "rollouts_to_step_cap": caps, "rollouts_stopped_infeasible": fails}
    if world == 1:
        res["solves_by_variant"] = {str(v): int((o["variant"] == v).sum()) for v in (4, 6, 8)}
        res["mean_ipm_iters"] = float(o["iters"][o["variant"] > 0].mean())
        if classify and fails:
            # genuine / solver split of the rollouts that stopped (reference: `break`, src/closed_loop.py:401-413): every one of
            # them replayed on the host and its last solve handed to an independent solver (tests/independent.py)
            try:
                from tests import independent as ind
                stopped = np.flatnonzero(o["flags"] == 3)
                n_stopped = len(stopped)
                if classify_max is not None and n_stopped > classify_max:      # bounded sample: one job per process, one wave
                    stopped = stopped[:classify_max]
                t1 = time.time()
                rows = ind.pool_map(ind.classify_stopped_world, [(first + int(i), n_dyn, int(o["steps"][i])) for i in stopped],
                                    max(1, min(192, os.cpu_count() or 1)))
                same = [r for r in rows if not r["replay_differs"]]
                res["stopped_infeasible_split"] = {
                    "sample": "%d of the %d stopped rollouts (lowest world indices%s)" % (len(stopped), n_stopped, "" if len(stopped) == n_stopped else "; --classify-all for every one"),
                    "classified": len(same), "feasible_point_exists_solver_failure": int(sum(r["feasible_point_found"] for r in same)),
                    "no_feasible_point_found": int(sum(not r["feasible_point_found"] for r in same)),
                    "host_replay_stops_elsewhere": len(rows) - len(same),
                    "solver_failures_world_step_variant_status": [list(r["tag"]) for r in same if r["feasible_point_found"]],
                    "method": "host replay of each stopped rollout (structured core), last solve to SciPy SLSQP from seven starts on the pinned model (window, straight line, last iterate, window moved 1.5 / 3 m to either side), %.1f s" % (time.time() - t1)}
            except Exception as e:          # noqa: BLE001
                res["stopped_infeasible_split"] = {"error": repr(e)}
    return res


def launch_ranks(n):
    """`python bench.py --gpus N` outside any launcher: re-run this command as N ranks, one per GPU of this node, under
    torch.distributed.run (rendezvous on 127.0.0.1, a free port) -- the same command line the driver would write by hand.
    Fails loudly (non-zero, no JSON line) when the node has fewer than N GPUs: a line that says n_gpus 1 must never come
    out of a --gpus N call."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node -- not running (no line printed)\n" % (n, have))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8192, help="instances per GPU")
    ap.add_argument("--horizon", type=int, default=5)
    ap.add_argument("--three-boxes", action="store_true", help="M=12 sub-config (3 boxes) instead of walls+box (M=6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--classify-all", action="store_true",
                    help="classify EVERY stopped C5 rollout on the host (about three minutes on 256 cores) instead of the first 48")
    ap.add_argument("--closed-loop-rollouts", type=int, default=4096,
                    help="config C5 reported beside the headline number at N=1 (0 = skip)")
    ap.add_argument("--full", action="store_true",
                    help="every secondary leg (dense-oracle timing, SLSQP cross-checks, the reference-picture replays, the x0-first re-runs of "
                         "C3 / open loop / C5, failure classification): about ten minutes.  The default run (about 100 s) keeps the headline, batch_1024, "
                         "cpu_baseline, roofline, config_c3, closed_loop (C5, 16 stopped rollouts classified), the start-order legs on the headline "
                         "batch, open_loop, the reference-picture replays and the static closed loops, and stops adding legs once "
                         "--budget-seconds of wall time are spent")
    ap.add_argument("--budget-seconds", type=float, default=150.0,
                    help="default run only: secondary legs that would start after this much wall time are skipped (and say so)")
    args = ap.parse_args()
    t_start = time.time()

    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))            # a bare `bench.py --gpus N` starts its own N ranks (one per GPU, RCCL)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks -- the line's n_gpus is the number of ranks, "
                 "so the two must agree" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:        # under torchrun the RCCL path runs even at world size 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    if not os.path.exists(_lib.LIB_PATH):               # build product, not in git: compile it (rank 0 first)
        import __graft_entry__ as ge
        if local_rank == 0:
            ge.build()
        if dist:
            dist.barrier()

    B, N = args.batch, args.horizon
    batch = sc.make_batch(B, N, three_boxes=args.three_boxes, first=rank * B,             # shard: instances rank*B ..
                          procs=max(1, min(16, (os.cpu_count() or 1) // max(world, 1))))
    M = sum(batch["m"])
    solver = BatchSolver(N, batch["m"], max_batch=B, device=dev)
    solver_rows = 3 + 3 * N + 3 + 2 * (N + 1) + 4 * N + 2 + (N + 1) * (2 * len(batch["m"]) + M + 4 * len(batch["m"]))
    prm = SolverParams()
    dv = {k: torch.as_tensor(batch[k], device=dev) for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    torch.cuda.synchronize()

    def step(out=None):
        return solver.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"],
                            prm, out=out)

    out = None
    for _ in range(args.warmup):
        out = step(out)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()                       # same stream the kernel is launched on (torch's current stream)
        out = step(out)
        ev[i][1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # configs[1] of BASELINE.json names the same workload at batch 1024: one wave per SIMD slot, so a launch lasts as
    # long as its slowest instance -- reported beside the headline (batch 8192, the north-star's size)
    small = None
    if world == 1 and B >= 1024:
        dv_s = {k: v[:1024].contiguous() for k, v in dv.items()}
        s_small = BatchSolver(N, batch["m"], max_batch=1024, device=dev)
        o_small = None
        for _ in range(2):
            o_small = s_small.solve(dv_s["variant"], dv_s["x0"], dv_s["u0"], dv_s["xref"], dv_s["A"], dv_s["b"], dv_s["Ts"],
                                    dv_s["term"], prm, out=o_small)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            o_small = s_small.solve(dv_s["variant"], dv_s["x0"], dv_s["u0"], dv_s["xref"], dv_s["A"], dv_s["b"], dv_s["Ts"],
                                    dv_s["term"], prm, out=o_small)
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t1
        ok_s = int(((o_small.status == 0) | (o_small.status == 1)).sum())
        small = {"workload": "C2 at batch 1024 (BASELINE configs[1])", "value": ok_s * args.steps / dt_s,
                 "unit": "MPC steps/s", "ms_per_step": dt_s / args.steps * 1e3, "success_rate": ok_s / 1024.0}
        s_small.close()

    ok = ((out.status == 0) | (out.status == 1)).sum().to(torch.float64)
    stats = torch.stack([torch.tensor(elapsed, dtype=torch.float64, device=dev), ok,
                         out.iters.to(torch.float64).sum(), out.info[:, 3].sum()])
    if dist:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        elapsed = float(tmax.item())
        from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.sharding import gather_outputs
        full = gather_outputs(dist, {"xopt": out.xopt, "uopt": out.uopt, "ts_opt": out.ts_opt, "status": out.status},
                              B * world, world)                            # the one data collective: outputs
        assert full["xopt"].shape[0] == B * world
    n_ok, it_sum, nf_sum = float(stats[1]), float(stats[2]), float(stats[3])
    total = B * world
    # the closed loop (config C5) sharded the same way: every rank runs the whole loop for its own rollouts
    c5_multi = None
    if dist is not None and args.closed_loop_rollouts > 0:
        try:
            c5_multi = closed_loop_c5(args.closed_loop_rollouts, first=rank * args.closed_loop_rollouts, dist=dist)
        except Exception as e:          # noqa: BLE001
            c5_multi = {"error": repr(e)}

    if rank == 0:
        value = n_ok * args.steps / elapsed             # instance-steps solved to converged/acceptable per second
        ab = alg_bytes(N, M)
        achieved = ab * B / (kern_ms * 1e-3) / 1e9
        kname = "obca_ipm_kernel_r4" if solver_rows <= 256 else "obca_ipm_kernel_r5" if solver_rows <= 320 else "obca_ipm_kernel_r6"
        if solver.specialised:          # the instantiation for this shape (csrc/obca_device.h: OBCA_SHAPES)
            kname = "obca_ipm_kernel_s%d_%d_%d" % (N, len(batch["m"]), M)
        pmc = pmc_summary(kname, B, N, M)
        live = pmc if (pmc and not pmc["stale"]) else {}
        line = {
            "metric": "OBCA MPC steps/sec (batch) at N=5, 3 obs", "value": value, "unit": "MPC steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2 generator (SURVEY 8d): demo1 corridor, 2 wall rows + 1 random box (M=%d), "
                                   "random start along an A*-like lattice path, obca_mpc4, default start ladder (reference window -> x0 -> zeros), seed %d"
                                   % (M, sc.SEED0),
                       "batch_per_gpu": B, "horizon_N": N, "obstacles": 3, "variant": "obca_mpc4",
                       "parallelism": "shard%d" % world},
            "attempted_steps_per_s": total * args.steps / elapsed, "success_rate": n_ok / total, "mean_ipm_iters": it_sum / total, "mean_kkt_factorisations": nf_sum / total,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         # HBM bytes per launch measured by the rocprofv3 --pmc passes committed under profiles/ (FETCH_SIZE x 2,
                         # the gfx950 correction of the microarchitecture guide, + WRITE_SIZE); null when no pass matches
                         "traffic": live.get("traffic_bytes"), "traffic_source": live.get("source"),
                         "traffic_kernel_source_hash": (pmc or {}).get("source_hash"), "traffic_git_sha": (pmc or {}).get("git_sha"),
                         "traffic_note": None if pmc is None else ("counter passes taken on OTHER kernel sources (hash %s, tree %s): not reported" % (pmc.get("source_hash"), kernel_source_hash()) if pmc["stale"] else "counter passes taken on the kernel sources in the tree"),
                         "valu_busy_frac": live.get("valu_busy_frac"), "wave_wait_frac": live.get("wave_wait_frac"),
                         # the bound that binds: VALU instructions issued x 64 lanes over what the chip's VALUs can issue in the
                         # launch time (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz; an fp64 instruction occupies its SIMD 4 cycles per wave)
                         "valu_issue_frac": (live["valu_insts"] * 64 / (live["launch_ms"] * 1e-3 * VALU_LANE_RATE)) if live.get("valu_insts") else None,
                         "kernel": kname, "kernel_ms": kern_ms, "algorithmic_bytes_per_instance": ab,
                         "fp64_model_frac": value / world * (nf_sum / total) * (N + 1) * (32 ** 3 / 3 + 2 * 32 ** 2) / 78.6e12
                         if M == 6 else None,
                         "note": "latency/fp64-VALU-issue bound by design (SURVEY 8d): ~1.3 KB of HBM traffic per solve, so the HBM "
                                 "fraction is ~3e-5 by construction; valu_issue_frac is the figure that describes the kernel.  "
                                 "fp64_model_frac = steps/s x KKT factorisations x (N+1)(s^3/3+2s^2), s=32, over 78.6 TF counts dense-block "
                                 "model flops the structured solve never executes -- a SURVEY 8d convention, not an achieved rate"},
        }
        if world == 8 and B == 8192:
            line["config"]["workload"] += " -- this is config C4 (65 536 scenarios sharded over 8 GPUs, gather only)"
        if c5_multi is not None:
            line["closed_loop"] = c5_multi
        if small is not None:
            line["batch_1024"] = small
        # ---- secondary legs.  (name, function, part of the default run).  A failure in one of them must not cost the headline line;
        # a default run skips the `--full` legs and whatever would start after the wall budget, and says so under the leg's key.
        legs = []
        cpu = dist is None and not args.no_cpu_baseline
        if cpu:
            legs.append(("cpu_baseline", lambda: cpu_baseline(batch, N), True))
            legs.append(("ipopt", lambda: ipopt_leg(batch, N), True))             # the reference's own solver, where it exists (not in this image)
            legs.append(("cpu_oracle", lambda: cpu_oracle(batch, N, min(args.cpu_seconds, 10.0)), False))
            legs.append(("independent_solver", lambda: independent_leg(solver, batch, dv, N, prm), False))   # SLSQP on the pinned model (tests/independent.py)
        if dist is None and args.closed_loop_rollouts > 0:
            def x0_first_all():
                r = start_order_leg(solver, dv, out, B, "x0")
                r["note"] = "obca_params.start_order = OBCA_START_X0_FIRST: x0 first for every variant -- for obca_mpc4 the default until obca_mpc 0.4, i.e. the configuration BENCH_r01 ... BENCH_r04 were measured in (NOT the default)"
                if args.full:
                    r["config_c3"] = config_c3(B, start_order="x0")
                    ol = open_loop(start_order="x0")
                    r["open_loop"] = {k: v for k, v in ol.items() if isinstance(v, dict) and "seconds" in v}
                    c5 = closed_loop_c5(args.closed_loop_rollouts, start_order="x0")
                    r["closed_loop"] = {k: c5[k] for k in ("value", "unit", "seconds", "converged_steps", "attempted_steps", "rollouts_to_step_cap", "rollouts_stopped_infeasible", "mean_ipm_iters") if k in c5}
                else:
                    r["config_c3"] = r["open_loop"] = r["closed_loop"] = {"skipped": "--full"}
                return r

            def zeros_first():
                r = start_order_leg(solver, dv, out, B, "zeros")
                r["note"] = "obca_params.start_order = OBCA_START_ZEROS_FIRST: the reference's literal all-zero start first (src/obca.py:856) -- the default of rounds 1-3, kept for comparison"
                return r
            def c5_dodge_off():
                c5 = closed_loop_c5(args.closed_loop_rollouts, dodge=False)
                r = {k: c5[k] for k in ("value", "unit", "seconds", "converged_steps", "attempted_steps", "rollouts_to_step_cap", "rollouts_stopped_infeasible", "mean_ipm_iters") if k in c5}
                r["note"] = ("obca_params.dodge = off (NOT the default): the ladder's last rung costs C5 a third of its launch and rescues ~20 of 4096 rollouts -- the launch lasts as long "
                             "as its longest ROLLOUT (a serial chain of <= 30 steps), and the rescued rollouts, which need the rung again at every later step, are the longest "
                             "(profiles/r06_c5_dodge.txt); the rung stays on because the reference's own demo11 run needs it (DESIGN.md section 2)")
                return r
            classify = cpu and args.full
            legs += [("config_c3", lambda: config_c3(B, classify=classify), True),
                     ("closed_loop", lambda: closed_loop_c5(args.closed_loop_rollouts, classify=cpu,
                                                            classify_max=None if args.classify_all else (48 if args.full else 16)), True),
                     ("x0_first", x0_first_all, True),
                     ("closed_loop_dodge_off", c5_dodge_off, True),
                     ("open_loop", lambda: open_loop(classify=classify), True),
                     ("reference_gif", reference_gif_leg, True),
                     ("zeros_first", zeros_first, True),
                     ("reference_report_figures", report_figures_leg, True),
                     # the same loop with the three static obstacles only, at the batch size BASELINE.json quotes
                     ("closed_loop_static", lambda: closed_loop_c5(B, n_dyn=0), True),
                     # optional extension, NOT reference behaviour (the reference cold-starts): shifted previous plan
                     ("closed_loop_static_warm_start", lambda: closed_loop_c5(B, n_dyn=0, warm_start=0.1), True)]
        leg_s = {}
        for name, fn, in_default in legs:
            if not args.full and not in_default:
                line[name] = {"skipped": "--full"}
                continue
            if not args.full and time.time() - t_start > args.budget_seconds:
                line[name] = {"skipped": "wall budget of the default run (%.0f s) spent; --full runs it" % args.budget_seconds}
                continue
            t_leg = time.time()
            try:
                line[name] = fn()
            except Exception as e:          # noqa: BLE001
                line[name] = {"value": None, "error": repr(e)} if name.startswith("cpu_") else {"error": repr(e)}
            leg_s[name] = round(time.time() - t_leg, 1)
        if cpu and line.get("ipopt") == "unavailable" and isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"]["note"] = "IPOPT unavailable (`import casadi` fails on this box); the reference publishes 3.7 s per solve at N=10 (src/simulation.py:231)"
        line["leg_seconds"] = leg_s
        line["wall_seconds"] = round(time.time() - t_start, 1)
        line["run"] = "full" if args.full else "default (secondary legs under a %.0f s wall budget; --full runs all of them)" % args.budget_seconds
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
