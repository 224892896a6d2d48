"""dev tool: throughput of ONE batch with one wavefront per instance against four wavefronts per instance (same body, same words) --
how much of the body's work more lanes can share.  Bounds what a two-wavefront (128-thread) variant could give the fused closed loop's
five-obstacle group: per instance at most 1 / (s + (1 - s) / w) faster with serial share s, on w times the hardware."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B, N = 8192, 5
for name, b in (("C2 (3 obstacles, 6 rows)", sc.make_batch(B, N)), ("three boxes (3 obstacles, 12 rows)", sc.make_batch(B, N, three_boxes=True))):
    res = {}
    for mode in ("wave", "multiwave"):
        s = BatchSolver(N, b["m"], max_batch=B)
        s.set_mode(mode)
        a = [b[k] for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")]
        o = s.solve(*a, SolverParams()); torch.cuda.synchronize()
        t = time.perf_counter(); o = s.solve(*a, SolverParams()); torch.cuda.synchronize(); res[mode] = time.perf_counter() - t
        s.close()
    w1, w4 = res["wave"], res["multiwave"]
    # per instance: 1024 slots of one wavefront against 256 slots of four -> time per instance t1 = w1 * 1024 / B, t4 = w4 * 256 / B
    sp4 = (w1 * 1024) / (w4 * 256)
    s_ser = (4.0 / sp4 - 1.0) / 3.0                      # Amdahl: 1 / sp4 = s + (1 - s) / 4
    sp2 = 1.0 / (s_ser + (1.0 - s_ser) / 2.0)
    print("%s: one wavefront per instance %.2f ms, four %.2f ms per %d solves -> an instance runs %.2f x faster on four wavefronts (serial share %.0f %%);"
          " two wavefronts: at most %.2f x per instance on twice the hardware = %.2f x the throughput" % (name, w1 * 1e3, w4 * 1e3, B, sp4, 100 * s_ser, sp2, sp2 / 2.0))
