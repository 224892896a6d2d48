"""dev tool: per-phase shader-clock shares of the solver (library built by tools/build_variant.sh prof -DOBCA_PROFILE):
python tools/gpu_prof.py B c2|c3|c3free N"""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import _lib, scenarios as sc
_lib.LIB_PATH = _lib.LIB_PATH.replace('libobca_mpc.so', 'libobca_mpc_prof.so')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
B=int(sys.argv[1]) if len(sys.argv)>1 else 768
N=int(sys.argv[3]) if len(sys.argv)>3 else 5
kind=sys.argv[2] if len(sys.argv)>2 else 'c2'
b=sc.make_batch_c3(B,N,gated=True) if kind=='c3' else sc.make_batch_c3(B,N,gated=False) if kind=='c3free' else sc.make_batch(B,N)
s=BatchSolver(N,b['m'],B)
prof=torch.zeros(B,20,dtype=torch.float64,device='cuda')
s.lib.obca_set_profile_buffer(s._h, ctypes.c_void_p(prof.data_ptr()))
for _ in range(2):
    out=s.solve(b['variant'],b['x0'],b['u0'],b['xref'],b['A'],b['b'],b['Ts'],b['term'],SolverParams())
torch.cuda.synchronize()
p=prof.cpu().numpy(); it=out.iters.cpu().numpy(); nf=out.info[:,3].cpu().numpy()
if p[:,19].max() > 0 and p[:,18].max() <= p[:,19].max():     # four-wavefront kernels: share of the solves whose sweep ran two-sided
    print('two-sided sweeps: %.3f of the solves (%.3f more with E^-1 in (1e6, 4e6])' % (p[:,18].sum()/p[:,19].sum(), p[:,10].sum()/p[:,19].sum()))
    p[:,18:] = 0; p[:,10] = 0
elif p[:,18].max() > 0:     # library built with -DOBCA_TWO_SIDED_CHECK: slots 18 / 19 hold step differences, not clocks
    print('two-sided vs one-sided sweep, same data: max rel. difference of the step %.2e (median of per-instance maxima %.2e), of the elastic multiplier steps %.2e (median %.2e)'
          % (p[:,18].max()/1e18, np.median(p[:,18])/1e18, p[:,19].max()/1e18, np.median(p[:,19])/1e18))
    p[:,18:] = 0
names=["grad+err","mu","rowE+gatherB","asm_stages","local","riccati","rowsteps","linesearch","accept","reeval","prologue","loop","r:FG+term","r:phaseA","r:phaseB","r:stage0","r:forward","r:recover","-","-"]
tot=p[:,:12].sum(1)
print('mean iters %.1f nfact %.1f total cycles/solve %.3e'%(it.mean(), nf.mean(), tot.mean()))
for i,n in enumerate(names):
    print('%-14s %6.2f%%  cycles/iter %9.0f'%(n, 100*p[:,i].sum()/tot.sum(), p[:,i].sum()/it.sum()))
st=out.status.cpu().numpy(); print('status counts', {int(k):int((st==k).sum()) for k in np.unique(st)})
