"""One line per kernel from `make -C sdfbox_amd/csrc resources` (hipcc -Rpass-analysis=kernel-resource-usage): SGPR / VGPR / spills /
scratch / occupancy / LDS.  Usage: python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(REPO, "sdfbox_amd", "csrc"), "resources"], capture_output=True, text=True).stderr
names = {}
cur = None
rows = []
for l in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", l)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
mangled = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"] + mangled, capture_output=True, text=True).stdout.splitlines()
print("# make -C sdfbox_amd/csrc resources (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage); scripts/kernel_resources.py")
print("# k_march / k_shadow / k_plain <CUR, COUNT, MODE>: CUR 0 generic, 1 cursor stack, 2 full-depth grid, 3 split grid; MODE 0 RGBA32F, 1 gamma RGBA8, 2 heat map, 3 wire")
for r, d in zip(rows, dem):
    d = d.replace("sdfhip::RenderParams", "RenderParams")
    print(f"{d:<60} SGPR {r['TotalSGPRs']:>3} VGPR {r['VGPRs']:>3} spill(s/v) {r['SGPRs Spill']}/{r['VGPRs Spill']} scratch {r['ScratchSize']} "
          f"occupancy {r['Occupancy']} LDS {r['LDS Size']}")
