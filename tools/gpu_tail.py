"""dev tool: how much of a launch is the tail, and what would a second, compacted launch for the ladder's retries buy?
One workgroup slot per SIMD (1024 on MI355X; 256 for the four-wavefront kernel) takes the instances in index order.  Service
times: KKT factorisations (closer to the cost than iterations), per PASS of the ladder -- measured by running every start of the
order alone (single_start) and the whole ladder.  Schedules simulated:
  ideal          sum / slots
  in-order       list schedule in index order, whole ladder per instance (what the hardware dispatcher does today)
  longest-first  list schedule, longest instance first (perfect foreknowledge)
  two launches   launch 1 = first start of every instance (in order), launch 2 = the failing instances' remaining passes, one
                 instance per slot (compacted; in order)
  speculative    launch 1 as above, launch 2 = every remaining pass of every failing instance as its OWN work item (all run
                 concurrently; the answer is the first feasible one in ladder order, so later passes may be wasted work)
python tools/gpu_tail.py > profiles/r05_tail_study.txt"""
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
for name, b, N, slots, first in (("C2 (one wavefront per instance, 1024 slots)", sc.make_batch(B, 5), 5, 1024, "x0"),
                                 ("C3 gated (four wavefronts per instance, 256 slots)", sc.make_batch_c3(B, 20, gated=True, procs=8), 20, 256, "window")):
    s = BatchSolver(N, b["m"], max_batch=B)
    args = (b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"])
    o = s.solve(*args, SolverParams())
    torch.cuda.synchronize()
    t = o.info[:, 3].cpu().numpy().astype(float)
    it = o.iters.cpu().numpy()
    o1 = s.solve(*args, SolverParams(start_order=first, single_start=True, dodge=False, max_iter_free=500 + 10 * N, max_iter_fixed=500 + 10 * N))
    torch.cuda.synchronize()
    t1 = o1.info[:, 3].cpu().numpy().astype(float)          # first start alone, capped at `patience`
    ok1 = np.isin(o1.status.cpu().numpy(), (0, 1))
    rest = np.maximum(t - t1, 0.0)[~ok1]                    # what the ladder spent after the first start on those that failed it
    ideal, inorder, lpt = t.sum() / slots, schedule(t, slots), schedule(np.sort(t)[::-1], slots)
    two = schedule(t1, slots) + (schedule(rest, slots) if len(rest) else 0.0)
    # speculative: a failing instance's remaining work split into its passes (at most 2 starts + 2 dodge passes; modelled as equal parts)
    parts = np.repeat(rest / 2.0, 2)
    spec = schedule(t1, slots) + (schedule(parts, slots) if len(parts) else 0.0)
    print("%s: factorisations mean %.0f max %.0f (iterations max %d); %d of %d fail the first start (their remaining passes: mean %.0f max %.0f)" %
          (name, t.mean(), t.max(), it.max(), (~ok1).sum(), B, rest.mean() if len(rest) else 0, rest.max() if len(rest) else 0))
    print("   ideal %.0f | in-order %.0f (+%.1f %%) | longest-first %.0f (+%.1f %%) | two launches %.0f (+%.1f %%) | speculative passes %.0f (+%.1f %%)" %
          (ideal, inorder, 100 * (inorder / ideal - 1), lpt, 100 * (lpt / ideal - 1), two, 100 * (two / ideal - 1), spec, 100 * (spec / ideal - 1)), flush=True)
    s.close()
