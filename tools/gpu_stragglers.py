"""dev tool: C3 gated half at B = 2048, one-sided against two-sided sweep: launch time, and the instances with the most
iterations (the ones that run into the 1000-iteration limit decide the launch time at this batch size)"""
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B,N=2048,20
b=sc.make_batch_c3(B,N,gated=True,procs=8)
for two in (False,True):
    s=BatchSolver(N,b["m"],max_batch=B); s.set_two_sided_sweep(two)
    for rep in range(2):
        torch.cuda.synchronize(); t=time.perf_counter()
        o=s.solve(b["variant"],b["x0"],b["u0"],b["xref"],b["A"],b["b"],b["Ts"],b["term"],SolverParams())
        torch.cuda.synchronize(); dt=time.perf_counter()-t
    it=o.iters.cpu().numpy(); st=o.status.cpu().numpy()
    top=np.argsort(it)[-6:]
    print("two_sided",two,"%.1f ms"%(dt*1e3),"mean it %.1f"%it.mean(),"top iters",it[top],"status",st[top],"sum it",it.sum(), flush=True)
