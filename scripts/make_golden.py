"""Writes the committed fixtures under tests/golden/:
  sphere_d4.asdf            the depth-4 sphere of cfg-1 as an .asdf file
  frames.npz                64x64 RGBA32F frames (alpha = step count) + counters
                            from the CPU oracle, 3 cameras x {sphere_d4, torus_d6},
                            plus one 48x32 4-spp path-traced frame per scene
  info_blocks.npz           the 112-byte Info block of each camera
Run from the repo root:  python scripts/make_golden.py
The frames are outputs of oracle/sdf_oracle.c, which restates the reference's
HLSL; the reference itself cannot run here -- "parity unpinned"."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle
import sdfbox_amd as sb
from conftest import CAMERAS, GOLDEN, make_camera

oracle.build(force=True)
scenes = {"sphere_d4": sb.sphere_d4(), "torus_d6": sb.torus_d6()}
scenes["sphere_d4"].Save(os.path.join(GOLDEN, "sphere_d4.asdf"))
frames, infos = {}, {}
for sname, od in scenes.items():
    for cname in CAMERAS:
        cam = make_camera(cname, 64, 64)
        img, cnt = oracle.render(od.Structs, od.Values, cam.State, 64, 64)
        frames[f"{sname}/{cname}/rgba"] = img
        frames[f"{sname}/{cname}/counters"] = cnt
        infos[cname] = np.frombuffer(bytes(cam.State), dtype=np.uint8)
        print(sname, cname, cnt, "lit", int((img[..., 0] > 0.0051).sum()), "sky", int((img[..., 2] == np.float32(0.2)).sum()))
# path-traced mode (BASELINE config 5 in small): 48x32, 4 spp, 3 bounces
for sname, od in scenes.items():
    cam = make_camera("default", 48, 32)
    img, cnt = oracle.render_pt(od.Structs, od.Values, cam.State, 48, 32, spp=4, max_bounces=3)
    frames[f"{sname}/path4/rgba"] = img
    frames[f"{sname}/path4/counters"] = cnt
    print(sname, "path4", cnt)
np.savez_compressed(os.path.join(GOLDEN, "frames.npz"), **frames)
np.savez_compressed(os.path.join(GOLDEN, "info_blocks.npz"), **infos)
