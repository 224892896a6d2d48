"""dev tool: soak test of the fused closed loop's work queue -- R batches of 2048 different worlds, every output word of the queue
schedules (rollouts handed from workgroup to workgroup every round: 2 = within the XCD, 1 = XCD to XCD) against one workgroup
per rollout; also reports where the workgroups ran (HW_REG_XCC_ID is read by the kernel itself)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.rollouts import DeviceRollouts, pack_worlds
from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.scenarios import make_world_c5
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bad = 0
for k in range(R):
    w = pack_worlds([make_world_c5(10000 * (k + 1) + i, n_dyn=2 if k % 2 == 0 else 1) for i in range(2048)])
    outs = []
    for env in ("2", "1", "0"):
        os.environ["OBCA_ROLLOUT_QUEUE"] = env
        dr = DeviceRollouts(w, N=5, warm_start=0.1) if k % 4 == 3 else DeviceRollouts(w, N=5)
        dr.run()
        outs.append({n: v.cpu().numpy() for n, v in dr.read().items()})
        torch.cuda.synchronize()
    diff = [(m, n) for m in (0, 1) for n in outs[m] if not np.array_equal(outs[m][n], outs[2][n])]
    bad += len(diff) > 0
    print("batch %d: steps %d, differing outputs: %s" % (k, int(outs[0]["steps"].sum()), diff or "none"), flush=True)
print("batches with differences:", bad)
