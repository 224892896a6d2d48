import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
Bs=[int(a) for a in sys.argv[1:]] or [1024, 8192]
base=sc.make_batch(512,5)
for B in Bs:
    rep=(B+511)//512
    b={k:(np.concatenate([v]*rep)[:B] if isinstance(v,np.ndarray) else v) for k,v in base.items()}
    res={}
    for mode in ('wave','lane'):
        s=BatchSolver(5,b['m'],B,mode=mode)
        dv={k:torch.as_tensor(b[k],device='cuda') for k in ('variant','x0','u0','xref','A','b','Ts','term')}
        out=None
        for i in range(2):
            torch.cuda.synchronize(); t=time.time()
            out=s.solve(dv['variant'],dv['x0'],dv['u0'],dv['xref'],dv['A'],dv['b'],dv['Ts'],dv['term'],SolverParams(),out=out)
            torch.cuda.synchronize(); dt=time.time()-t
        st=out.status.cpu().numpy()
        res[mode]=(out.xopt.cpu().numpy(), out.iters.cpu().numpy())
        print('B',B,mode,'%.1f ms'%(dt*1e3),'%.0f solves/s'%(B/dt),'ok frac',np.mean((st==0)|(st==1)),'iters',out.iters.float().mean().item(), flush=True)
        s.close()
    d=np.abs(res['wave'][0]-res['lane'][0]).max(axis=(1,2))
    print('   wave vs lane max|dx| median %.1e max %.1e; same iters %.2f'%(np.median(d), d.max(), np.mean(res['wave'][1]==res['lane'][1])))
