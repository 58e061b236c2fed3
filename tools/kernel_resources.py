#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of the engine as hipcc compiles it for gfx950 (`-Rpass-analysis=kernel-resource-usage`).
  python tools/kernel_resources.py [filter] [-- extra hipcc flags]     e.g.  tools/kernel_resources.py resident -DJSLP_NODE512_WAVES=8
Prints a markdown table (what profiles/rNN_kernel_resources.md holds)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
flt = args[0] if args else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
       "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kres.so",
       os.path.join(ROOT, "jslpsolver_amd", "csrc", "jslp_hip.hip")] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else "c++filt", n], capture_output=True, text=True).stdout.strip().replace("void ", "")
    except Exception:
        return n
print("| kernel | VGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | waves/SIMD | LDS B |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    name = re.sub(r"\(.*", "", demangle(r["name"]))
    if flt and flt not in name:
        continue
    print("| `%s` | %s | %s | %s | %s | %s | %s | %s |" % (name, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("VGPRs Spill"),
                                                      r.get("SGPRs Spill"), r.get("Occupancy"), r.get("LDS Size")))
