"""dev tool: how much of a launch is the tail?  One workgroup slot per SIMD (1024 on MI355X) takes the instances in index
order; with the measured iteration counts as service times: ideal = sum / slots, in-order list schedule (what the hardware
dispatcher does), longest-first schedule (what perfect foreknowledge would allow)."""
import sys, heapq
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams


def schedule(t, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for v in t:
        heapq.heappush(h, heapq.heappop(h) + v)
    return max(h)


B = 8192
for name, b, N, slots in (("C2 (one wavefront per instance, 1024 slots)", sc.make_batch(B, 5), 5, 1024),
                          ("C3 gated (four wavefronts per instance, 256 slots)", sc.make_batch_c3(B, 20, gated=True, procs=8), 20, 256)):
    s = BatchSolver(N, b["m"], max_batch=B)
    o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
    torch.cuda.synchronize()
    t = o.info[:, 3].cpu().numpy().astype(float)            # KKT factorisations: closer to the cost than the iteration count
    it = o.iters.cpu().numpy()
    ideal, inorder, lpt = t.sum() / slots, schedule(t, slots), schedule(np.sort(t)[::-1], slots)
    print("%s: factorisations mean %.0f max %.0f (iterations max %d) | ideal %.0f  in-order %.0f (+%.1f %%)  longest-first %.0f (+%.1f %%)" %
          (name, t.mean(), t.max(), it.max(), ideal, inorder, 100 * (inorder / ideal - 1), lpt, 100 * (lpt / ideal - 1)), flush=True)
    s.close()
