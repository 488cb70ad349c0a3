#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch / occupancy table of one HIP source, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.py rebvo_amd/csrc/stage_b.hip [name-filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
here = os.path.dirname(os.path.abspath(src))
root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       f"-I{root}/include", f"-I{here}", f"-I{root}/rebvo_amd/host/include", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + extra
if src.endswith("stage_imu.hip"):
    cmd += ["-mllvm", "-unroll-threshold=2000"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name|\s*Name): (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scr':>5} {'occ':>4} {'LDS':>7} {'sS':>4} {'vS':>4}  kernel")
for r in rows:
    name = demangle(r["name"])
    if filt and filt not in name:
        continue
    print(f"{r.get('vgpr', -1):5d} {r.get('agpr', -1):5d} {r.get('sgpr', -1):5d} {r.get('scratch', -1):5d} {r.get('occ', -1):4d} {r.get('lds', -1):7d} "
          f"{r.get('sspill', -1):4d} {r.get('vspill', -1):4d}  {name[:110]}")
