"""Registers, spills, scratch and occupancy of every kernel of a translation unit, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU): tools/kernel_resources.py den_lazy.hip [extra flags]"""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "pychain_amd", "csrc", sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-I", os.path.join(REPO, "include"),
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"remark: (Function Name|Name): (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"pychain_hip::\(anonymous namespace\)::|\(anonymous namespace\)::|pychain_hip::", "", name)
        name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
print("%-64s %5s %6s %7s %7s %4s" % ("kernel", "VGPR", "vspill", "sspill", "scratch", "occ"))
for r in rows:
    print("%-64s %5s %6s %7s %7s %4s" % (r["name"][:64], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
                                          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
