"""Launches the kernels the profiles/ passes look at, and nothing else: `c2` = obca_ipm_kernel_r4 at B = 8192 (headline),
`c5` = obca_rollout_fused_kernel_r5 (4096 rollouts, two moving boxes), `c3g` = obca_ipm_kernel_mw_r5 (N = 20, gated).
Run under `rocprofv3 --kernel-trace --stats` or one `--pmc` pass at a time (tools/profile.sh)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import os
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib
if os.environ.get("OBCA_LIB"):          # a dev build next to the product library (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.join(_lib.HERE, os.environ["OBCA_LIB"])
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams

what = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = None
if what == "c2":
    B, N = 8192, 5
    b = sc.make_batch(B, N)
elif what == "c3g":
    B, N = 2048, 20
    b = sc.make_batch_c3(B, N, gated=True, procs=16)
if what in ("c2", "c3g"):
    s = BatchSolver(N, b["m"], max_batch=B, mode=mode)
    dv = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    out = None
    for _ in range(reps):
        out = s.solve(dv["variant"], dv["x0"], dv["u0"], dv["xref"], dv["A"], dv["b"], dv["Ts"], dv["term"], SolverParams(), out=out)
    torch.cuda.synchronize()
    print(what, "ok", float(((out.status == 0) | (out.status == 1)).float().mean()), "iters", float(out.iters.float().mean()))
else:
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
    w = pack_worlds([sc.make_world_c5(i, n_dyn=2) for i in range(4096)])
    dr = DeviceRollouts(w, N=5)
    for _ in range(max(1, reps - 2)):
        dr.reset(); dr.run(); torch.cuda.synchronize()
    o = dr.read()
    print("c5 converged steps", int(o["steps"].sum()))
