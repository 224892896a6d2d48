"""Condenses the rocprofv3 outputs of tools/profile.sh (gpurun_out/prof_<tag>/) into the files committed under profiles/:
per-kernel launch durations from the kernel trace, the --stats table, FETCH_SIZE / WRITE_SIZE per launch (KiB as rocprofv3
reports them; FETCH_SIZE doubled for HBM bytes as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950) and the SQ
issue counters, plus profiles/<tag>_pmc_summary.json, which bench.py reads for `roofline.traffic`."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
SHAPES = {"c2": dict(B=8192, N=5, M=6), "c5": dict(B=4096, N=5, M=14), "c3g": dict(B=2048, N=20, M=14)}


def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return hits[0] if hits else None


sys.path.insert(0, ROOT)
import subprocess  # noqa: E402
import bench  # noqa: E402  (kernel_source_hash: the sources these passes measured -- run this right after the profile run)
try:
    sha = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
except Exception:
    sha = None
out = {"tag": tag, "kernel_source_hash": bench.kernel_source_hash(), "git_sha": sha, "kernels": []}
# stats + per-launch durations of the bench run
st = find(os.path.join(src, "trace"), "*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(dst, tag + "_kernel_stats.csv"))
kt = find(os.path.join(src, "trace"), "*kernel_trace.csv")
if kt:
    with open(os.path.join(dst, tag + "_solver_launches_from_kernel_trace.csv"), "w") as f:
        f.write("Kernel_Name,Grid_Size,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Duration_ms\n")
        for r in rows(kt):
            if "obca" in r["Kernel_Name"]:
                f.write("%s,%s,%s,%s,%s,%s,%s,%s,%.4f\n" % (r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")),
                        r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""),
                        r.get("Scratch_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", ""),
                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
for name in ("bench_under_rocprof.json",):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, tag + "_" + name))
for tgt, shape in SHAPES.items():
    per = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_%s_*" % tgt))):
        if not os.path.isdir(d):
            continue
        cc = find(d, "*counter_collection.csv")
        if not cc:
            continue
        for r in rows(cc):
            k = r["Kernel_Name"]
            if "obca" not in k:
                continue
            e = per.setdefault(k, {})
            e.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            e["_dur"] = e.get("_dur", []) + [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6]
            e["_meta"] = {x: r.get(x) for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
        shutil.copy(cc, os.path.join(dst, "%s_pmc_%s_%s.csv" % (tag, tgt, os.path.basename(d).split("_", 2)[2][:24])))
    for k, e in per.items():
        main = max(e["_dur"]) > 1.0                       # skip the tiny helper kernels
        if not main:
            continue
        mean = lambda v: sum(v) / len(v)
        big = lambda name: [v for v, d_ in zip(e.get(name, []), e["_dur"]) if True]
        full = [d_ for d_ in e["_dur"] if d_ >= 0.5 * max(e["_dur"])]     # full-size launches (not warm-ups, not the fused loop's clean-up pass)
        rec = dict(kernel=k, target=tgt, **shape, meta=e["_meta"], launch_ms=mean(full))
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            fk, wk = max(e["FETCH_SIZE"]), max(e["WRITE_SIZE"])        # the full-size launches (warm-up launches are smaller)
            rec.update(fetch_size_kib=fk, write_size_kib=wk, traffic_bytes=2 * fk * 1024 + wk * 1024,
                       source=["profiles/%s_pmc_%s_FETCH_SIZE.csv" % (tag, tgt), "profiles/%s_pmc_%s_WRITE_SIZE.csv" % (tag, tgt)])
        if "SQ_WAVE_CYCLES" in e:
            wc = max(e["SQ_WAVE_CYCLES"])
            g = lambda n: max(e[n]) if n in e else None
            rec.update(sq_wave_cycles=wc, valu_busy_frac=g("SQ_ACTIVE_INST_VALU") / wc, wave_wait_frac=g("SQ_WAIT_ANY") / wc,
                       inst_any_frac=g("SQ_ACTIVE_INST_ANY") / wc, issue_stall_frac=(g("SQ_WAIT_INST_ANY") or 0) / wc,
                       valu_insts_per_wave=g("SQ_INSTS_VALU") / max(g("SQ_WAVES") or 1, 1), valu_insts=g("SQ_INSTS_VALU"))
        out["kernels"].append(rec)
with open(os.path.join(dst, tag + "_pmc_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1)[:3000])
