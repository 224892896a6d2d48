#!/usr/bin/env python3
"""Drop-in use on an MI355X: the reference's demo driven through this package (needs a ROCm GPU).

    python examples/closed_loop_demo.py demo8            # one rollout, the reference's closedLoop call surface
    python examples/closed_loop_demo.py --monte-carlo 4096   # device-resident Monte-Carlo rollouts (config C5)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("demo", nargs="?", default="demo8")
    ap.add_argument("--monte-carlo", type=int, default=0, help="number of seeded C5 worlds instead of one demo")
    ap.add_argument("--warm-start", action="store_true", help="optional extension, not reference behaviour")
    ap.add_argument("--start-order", choices=("default", "x0", "window", "zeros"), default="default",
                    help="obca_params.start_order (include/obca_mpc.h): default = x0 first for obca_mpc4, the reference window first for obca_mpc6 / 8; "
                         "or that start first for every variant (zeros = the reference's literal all-zero start)")
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    import numpy as np
    import torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.closed_loop import closedLoop
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.demo_setting import problemSetting
    if not args.monte_carlo:
        cl = closedLoop(problemSetting(args.demo))          # same attributes / methods as the reference class
        cl.obca_solver.start_order = args.start_order
        t0 = time.time()
        x_open, x_closed, u_closed, T_closed = cl.closed_loop_mpc4()
        print("%s: %d closed-loop steps in %.2f s, final pose %s, step lengths %s" % (
            args.demo, len(u_closed), time.time() - t0, np.round(x_closed[-1], 3), np.round(T_closed[:5], 3)))
        return
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds, reference_lists
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
    w = pack_worlds([make_world_c5(i) for i in range(args.monte_carlo)])
    prm = None
    if args.start_order != "default":
        from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
        prm = SolverParams(xL=getattr(w, "xL", (0.0, 0.0)), xU=getattr(w, "xU", (39.0, 10.0)), start_order=args.start_order)
    dr = DeviceRollouts(w, N=5, warm_start=0.1 if args.warm_start else None, params=prm)
    torch.cuda.synchronize()
    t0 = time.time()
    out = dr.run().read()
    torch.cuda.synchronize()
    dt = time.time() - t0
    steps, flags = out["steps"].cpu().numpy(), out["flags"].cpu().numpy()
    print("%d rollouts, %d converged closed-loop steps in %.2f s (%.0f steps/s); ended at goal/cap/failed: %s" % (
        args.monte_carlo, steps.sum(), dt, steps.sum() / dt, np.bincount(flags, minlength=4)[1:].tolist()))
    first = reference_lists(out, w, 0)                     # the lists the reference's plot routine takes
    print("rollout 0: %d steps, last pose %s" % (len(first["u_closed"]), np.round(first["x_closed"][-1], 3)))


if __name__ == "__main__":
    main()
