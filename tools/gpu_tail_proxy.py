"""dev tool: can a cheap a-priori proxy order the instances well enough to shorten the tail?  Proxy = clearance of the
reference window from the obstacles (min over stages and obstacles of the polytope's signed distance max_j (a_j . p - b_j))."""
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
for name, b, N, slots in (("C2", sc.make_batch(B, 5), 5, 1024), ("C3 free", sc.make_batch_c3(B, 20, gated=False, procs=8), 20, 256),
                          ("C3 gated", sc.make_batch_c3(B, 20, gated=True, procs=8), 20, 256)):
    s = BatchSolver(N, b["m"], max_batch=B)
    o = s.solve(b["variant"], b["x0"], b["u0"], b["xref"], b["A"], b["b"], b["Ts"], b["term"], SolverParams())
    torch.cuda.synchronize()
    t = o.info[:, 3].cpu().numpy().astype(float)
    s.close()
    A, bb, xr, x0 = b["A"], b["b"], b["xref"], b["x0"]           # [B,N+1,M,2], [B,N+1,M], [B,3,N+1]
    m = b["m"]
    off = np.concatenate([[0], np.cumsum(m)])
    p = np.transpose(xr[:, :2, :], (0, 2, 1))                    # [B,N+1,2]
    sd = np.einsum("bkmj,bkj->bkm", A, p) - bb                   # [B,N+1,M]
    clear = np.full(B, np.inf)
    for i in range(len(m)):
        if m[i] >= 3:                                            # closed polytopes only (walls are half-planes)
            d = sd[:, :, off[i]:off[i + 1]].max(-1)              # signed distance-like, per stage
            clear = np.minimum(clear, d.min(-1))
    dist0 = np.linalg.norm(x0[:, :2] - xr[:, :2, 0], axis=1)
    ideal, inorder, lpt = t.sum() / slots, schedule(t, slots), schedule(np.sort(t)[::-1], slots)
    line = "%-8s ideal %.0f in-order +%.1f %% longest-first +%.1f %%" % (name, ideal, 100 * (inorder / ideal - 1), 100 * (lpt / ideal - 1))
    for pname, key in (("clearance", clear), ("start offset", -dist0), ("clearance - offset", clear - dist0)):
        order = np.argsort(key, kind="stable")
        line += " | by %s +%.1f %% (corr %.2f)" % (pname, 100 * (schedule(t[order], slots) / ideal - 1), np.corrcoef(key, t)[0, 1])
    print(line, flush=True)
