"""dev tool: the C5 closed loop (4096 rollouts, two moving boxes) under the three schedules of the fused kernel
(OBCA_ROLLOUT_QUEUE = 2 one queue per XCD, 1 one global queue, 0 one workgroup per rollout): time, and every output word equal?"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc
if os.environ.get("OBCA_LIB"):      # an alternative build of the library (file name inside the package)
    _lib.LIB_PATH = os.path.join(_lib.HERE, os.environ["OBCA_LIB"])
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
w = pack_worlds([sc.make_world_c5(i, n_dyn=2) for i in range(4096)])
ref = None
for env in (os.environ.get("OBCA_QUEUE_MODES") or "2,1,0,2,1").split(","):
    os.environ["OBCA_ROLLOUT_QUEUE"] = env
    dr = DeviceRollouts(w, N=5)
    dr.run(1); torch.cuda.synchronize(); dr.reset(); torch.cuda.synchronize()
    t = time.perf_counter(); dr.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    o = {k: v.cpu().numpy() for k, v in dr.read().items()}
    if ref is None: ref = o
    same = all(np.array_equal(o[k], ref[k]) for k in o)
    print("queue mode %s: %.4f s, %d converged steps -> %.0f steps/s, equal to first: %s" % (env, dt, o["steps"].sum(), o["steps"].sum() / dt, same), flush=True)
