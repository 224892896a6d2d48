import sys, numpy as np, torch
sys.path.insert(0, ".")
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import SolverParams
B=4096
worlds=[sc.make_world_c5(i, n_dyn=2) for i in range(B)]
res={}
for name,kw in (("on",{}),("off",dict(dodge=False))):
    w=pack_worlds(worlds)
    dr=DeviceRollouts(w,N=5,params=SolverParams(xL=getattr(w,"xL",(0.0,0.0)),xU=getattr(w,"xU",(39.0,10.0)),**kw))
    dr.run(); torch.cuda.synchronize()
    res[name]={k:v.cpu().numpy() for k,v in dr.read().items()}
on,off=res["on"],res["off"]
it=on["iters"].astype(np.int64)*(on["variant"]>0); tot=it.sum(1)
ito=off["iters"].astype(np.int64)*(off["variant"]>0); toto=ito.sum(1)
order=np.argsort(tot)[::-1][:12]
print("longest rollouts with the rung on: world, iterations on/off, steps on/off, flags on/off")
for b in order: print(int(b), int(tot[b]), int(toto[b]), int(on["steps"][b]), int(off["steps"][b]), int(on["flags"][b]), int(off["flags"][b]))
b=int(order[0])
print("world",b,"per step (variant,status,iters) on:", [(int(on["variant"][b,k]),int(on["status"][b,k]),int(on["iters"][b,k])) for k in range(30)])
print("world",b,"per step off:", [(int(off["variant"][b,k]),int(off["status"][b,k]),int(off["iters"][b,k])) for k in range(30)])
print("histogram of rollout totals (on):", np.percentile(tot,[50,90,99,99.9,100]).tolist(), "(off):", np.percentile(toto,[50,90,99,99.9,100]).tolist())
d=tot-toto; print("rollouts with more iterations with the rung:", int((d>0).sum()), "sum of extra", int(d[d>0].sum()), "top extras", np.sort(d)[::-1][:10].tolist())
