"""dev tool (library built with OBCA_HIPCC_FLAGS=-DOBCA_RO_STATS): where do the persistent workgroups of the fused closed-loop
kernel spend their time -- waiting for an item, working -- and when do they finish?"""
import sys, ctypes, time
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
B = 4096
w = pack_worlds([sc.make_world_c5(i, n_dyn=2) for i in range(B)])
dr = DeviceRollouts(w, N=5)
dr.run(); torch.cuda.synchronize()
dr.reset(); torch.cuda.synchronize(); t = time.perf_counter(); dr.run(); torch.cuda.synchronize(); print("wall %.3f s" % (time.perf_counter() - t))
lib = dr.lib if hasattr(dr, "lib") else dr._lib
h = dr._h if hasattr(dr, "_h") else dr.h
out = (ctypes.c_int32 * (4 * 1024))()
lib.obca_rollouts_debug_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
print("rc", lib.obca_rollouts_debug_stats(h, out, 4 * 1024))
a = np.frombuffer(out, dtype=np.int32).reshape(1024, 4).astype(float)
wait, work, items, tend = a[:, 0] * 1e-8, a[:, 1] * 1e-8, a[:, 2], a[:, 3]
tend = (tend - tend.min()) * 1e-8
print("per workgroup: wait mean %.3f s (max %.3f), work mean %.3f s (min %.3f max %.3f), items mean %.0f; finish spread %.3f s" %
      (wait.mean(), wait.max(), work.mean(), work.min(), work.max(), items.mean(), tend.max()))
