"""MI355X-native batched OBCA-MPC solver (hot path of the reference's src/obca.py + src/closed_loop.py).

    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.obca import obca          # drop-in class
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver  # batched API
"""
__all__ = ["obca", "solver", "_lib"]
