"""dev tool: registers, scratch and LDS of every kernel in the built objects (from the code-object metadata), e.g. to
check after a change that the hot kernels are still spill-free.   python tools/kernel_resources.py [obj_dir]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(obj):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + os.path.join(tmp, "fb.bin"), obj],
                           stderr=subprocess.DEVNULL)
        if r.returncode != 0:
            return out                       # no device code in this object
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + os.path.join(tmp, "fb.bin"), "--output=" + os.path.join(tmp, "dev.co"), "--unbundle"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, "dev.co")], capture_output=True,
                             text=True).stdout
        size = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", os.path.join(tmp, "dev.co")], capture_output=True, text=True).stdout
    sizes = {m.group(2): int(m.group(1)) for m in re.finditer(r"\s+\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(\S+)", size)}
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        out[name] = dict(vgpr=int(g("vgpr_count").group(1)), agpr=int(blk.split()[0]), sgpr=int(g("sgpr_count").group(1)),
                         scratch=int(g("private_segment_fixed_size").group(1)), lds_static=int(g("group_segment_fixed_size").group(1)),
                         code_bytes=sizes.get(name, 0))
    return out


if __name__ == "__main__":
    d = sys.argv[1] if len(sys.argv) > 1 else None
    if d is None:
        stamp = os.path.join(ROOT, "vehicle_motion_planning_with_obstacles_avoidance_using_mpc_amd", "csrc", "_build", "linked_flags")
        d = open(stamp).read().strip()
    for obj in sorted(glob.glob(os.path.join(d, "*.o"))):
        for k, v in sorted(resources(obj).items()):
            print("%-18s %-36s vgpr %3d agpr %3d sgpr %3d scratch %5d B  static LDS %5d B  code %6d B" %
                  (os.path.basename(obj), k, v["vgpr"], v["agpr"], v["sgpr"], v["scratch"], v["lds_static"], v["code_bytes"]))
