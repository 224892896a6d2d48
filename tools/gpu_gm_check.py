"""dev tool: the kernel for shapes beyond the LDS (mode "global": four wavefronts per instance, rows in an HBM workspace)
against the LDS-resident four-wavefront kernel on shapes both run, and its time on the reference's long-horizon open-loop plans"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams


def run(b, N, mode):
    B = len(b["variant"])
    s = BatchSolver(N, b["m"], max_batch=B, mode=mode)
    dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    o = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams())
    torch.cuda.synchronize()
    t = time.perf_counter()
    o = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams(), out=o)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    r = dict(x=o.xopt.cpu().numpy(), st=o.status.cpu().numpy(), it=o.iters.cpu().numpy(), dt=dt)
    s.close()
    return r


for name, b, N in (("C2 256", sc.make_batch(256, 5), 5), ("C3 free 256", sc.make_batch_c3(256, 20, gated=False, procs=8), 20),
                   ("C3 gated 256", sc.make_batch_c3(256, 20, gated=True, procs=8), 20)):
    a, g = run(b, N, "multiwave"), run(b, N, "global")
    print("%s: status equal %s, iterations equal %s, x bit-identical %s | multiwave %.1f ms, global %.1f ms" %
          (name, np.array_equal(a["st"], g["st"]), np.array_equal(a["it"], g["it"]), np.array_equal(a["x"], g["x"]), a["dt"] * 1e3, g["dt"] * 1e3), flush=True)

from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca
for demo, N in (("demo1", 10), ("demo1", 40), ("demo1", 74), ("demo9", 40), ("demo9", 74)):
    s = obca()
    cl = closedLoop(problemSetting(demo), solver=s)
    cl.N_free = N
    cl.mpc_openLoop_freeTime()
    torch.cuda.synchronize()
    t = time.perf_counter()
    cl.mpc_openLoop_freeTime()
    torch.cuda.synchronize()
    print("%s N=%d: feas %s Ts_opt %.5f, %.3f s" % (demo, N, cl.feas, cl.Ts_opt, time.perf_counter() - t), flush=True)
