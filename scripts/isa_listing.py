"""Emit the loops of one kernel as hipcc builds them: compiles render.hip to gfx950 assembly (device only) and prints every
"Depth=1" loop of the kernel (header block to the block after its last one) with static instruction counts.
Usage: python scripts/isa_listing.py [mangled-kernel-name]   (default: k_march<CUR_STACK_SPLIT, false, OUT_RGBA32F, false>, the bench
frame's kernel: loop 1 = primary march, loop 2 = shadow march)
The committed profiles/r05_isa_k_march.txt is this output under a hand-written header holding the dynamic counts."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sdfbox_amd", "csrc")
KERNEL = sys.argv[1] if len(sys.argv) > 1 else "_ZN6sdfhip7k_marchILi3ELb0ELi0ELb0EEEvNS_12RenderParamsE"

with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "dev.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                    "-fvisibility=hidden", "--offload-device-only", "-S", "render.hip", "-o", out],
                   cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read().splitlines()

start = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ":"))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
body = text[start:end]

# the loops of the kernel: the blocks that name "in Loop: Header=BBn_m Depth=1" (or are that header), grouped by header, each
# from its first block to the block after its last one
labels = [i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l) or re.match(r"; %bb\.\d+:", l)]
loops = {}
for n, i in enumerate(labels):
    m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=1", body[i])
    h = m.group(1) if m else None
    if h is None and "Inner Loop Header: Depth=1" in body[i]:
        h = re.match(r"\.L(BB\d+_\d+):", body[i]).group(1)
    if h is not None:
        stop = labels[n + 1] if n + 1 < len(labels) else len(body)
        first, _ = loops.get(h, (i, None))
        loops[h] = (first, stop)

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

res = {}
for l in text[end:end + 60]:
    m = re.match(r";\s*(NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize|codeLenInByte)[^:]*:\s*(\d+)", l.strip())
    if m:
        res[m.group(1)] = int(m.group(2))
print(f"# {KERNEL}")
print(f"# kernel: {res}")
for n, (h, (first, stop)) in enumerate(sorted(loops.items(), key=lambda kv: kv[1][0])):
    loop = body[first:stop]
    counts = {}
    for l in loop:
        k = kind(l)
        if k:
            counts[k] = counts.get(k, 0) + 1
    print()
    print(f"# ---- loop {n + 1} (header {h}); static instructions, all paths: {counts.get('valu', 0)} VALU, {counts.get('salu', 0)} SALU, "
          f"{counts.get('vmem_load', 0)} global loads, {counts.get('vmem_store', 0)} stores/atomics, {counts.get('lds', 0)} LDS")
    print("\n".join(loop))
