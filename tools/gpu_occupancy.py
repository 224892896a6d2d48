"""dev tool: the headline launch at forced occupancies (OBCA_LDS_PAD pads the one-wavefront kernel's LDS request so that only
k workgroups fit a CU's 160 KB) -- how much of the launch is latency a second wave per SIMD would hide?
    python tools/gpu_occupancy.py [B]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd import scenarios as sc
    from vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd.solver import BatchSolver, SolverParams
    B = int(sys.argv[2])
    b = sc.make_batch(B, 5, procs=16)
    s = BatchSolver(5, b["m"], max_batch=B)
    d = {k: torch.as_tensor(b[k], device="cuda") for k in ("variant", "x0", "u0", "xref", "A", "b", "Ts", "term")}
    o = None
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = s.solve(d["variant"], d["x0"], d["u0"], d["xref"], d["A"], d["b"], d["Ts"], d["term"], SolverParams(), out=o); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(json.dumps({"lds": s.lds_bytes, "ms": min(ts[1:]), "ok": int(((o.status == 0) | (o.status == 1)).sum())}))
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "8192"
base = None
for per_cu in (8, 7, 6, 5, 4, 3, 2):
    env = dict(os.environ)
    if base is not None:
        pad = 160 * 1024 // per_cu - base - 512
        if pad <= 0:
            print("%d per CU: fits already" % per_cu); continue
        env["OBCA_LDS_PAD"] = str(pad - pad % 8)
    r = subprocess.run([sys.executable, __file__, "--child", B], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(r.stderr[-500:]); break
    j = json.loads(line[-1])
    if base is None:
        base = j["lds"]
        print("no pad: lds %d B -> %d per CU; %.2f ms" % (base, 160 * 1024 // (base + 512), j["ms"]))
    else:
        print("%d per CU (pad %s): %.2f ms" % (per_cu, env.get("OBCA_LDS_PAD"), j["ms"]))
