"""Emit the hot loop of one kernel as hipcc builds it: compiles sdfhip_device.hip to gfx950 assembly (device only), finds the
kernel's innermost "Loop Header ... Depth=1" region up to its backward branch, and prints it with static instruction counts.
Usage: python scripts/isa_listing.py [mangled-kernel-name]   (default: k_march<CUR_STACK_FULL, false, OUT_RGBA32F>)
The committed profiles/r02_isa_k_march.txt is this output under a hand-written header holding the dynamic counts."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sdfbox_amd", "csrc")
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "_ZN6sdfhip7k_marchILi2ELb0ELi0EEEvNS_12RenderParamsE"

with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "dev.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                    "-fvisibility=hidden", "--offload-device-only", "-S", "sdfhip_device.hip", "-o", out],
                   cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()

start = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ":"))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
body = text[start:end]

# the march loop: the blocks that name "Loop Header=BBn_m Depth=1" or are that header, up to the last branch back to it
header = None
for l in body:
    m = re.search(r"Loop Header: Depth=1", l)
    if m:
        header = l
        break
labels = [i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)]
in_loop = [i for i in labels if "in Loop: Header=" in body[i + 1] or "Loop Header: Depth=1" in body[i] or "Loop Header: Depth=1" in body[i + 1]
           or "in Loop: Header=" in body[i]]
first = in_loop[0]
last_label = in_loop[-1]
stop = next(i for i in labels + [len(body)] if i > last_label) if last_label != labels[-1] else len(body)
loop = body[first:stop]

def kind(l):
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
        return None
    op = t[0]
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "global_atomic")):
        return "vmem_store"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    return "other"

counts = {}
for l in loop:
    k = kind(l)
    if k:
        counts[k] = counts.get(k, 0) + 1
res = {}
for l in text[end:end + 60]:
    m = re.match(r";\s*(NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize|codeLenInByte)[^:]*:\s*(\d+)", l.strip())
    if m:
        res[m.group(1)] = int(m.group(2))
print(f"# {KERNEL}")
print(f"# static instructions in the loop (all paths): {counts.get('valu', 0)} VALU, {counts.get('salu', 0)} SALU, "
      f"{counts.get('vmem_load', 0)} global loads, {counts.get('vmem_store', 0)} stores/atomics, {counts.get('lds', 0)} LDS")
print(f"# kernel: {res}")
print()
print("\n".join(loop))
