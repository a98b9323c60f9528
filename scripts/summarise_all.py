"""python scripts/summarise_all.py r02  -- condense every gpurun_out/prof_<round>_* into profiles/ (see summarise_profile.py)."""
import subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
G = "grid8+blocks"          # the default grid of both stand-in scenes: coarse level 8 + blocks (bench.py's key names it)
for tag, key in [("1080p", f"1920x1080:dragon_standin_d9:default:{G}"), ("1080p_orbit", f"1920x1080:dragon_standin_d9:default:orbit90:{G}"), ("1080p_onekernel", f"1920x1080:dragon_standin_d9:one-kernel:{G}"),
                 ("1080p_queue", f"1920x1080:dragon_standin_d9:default:shadow-queue:{G}"), ("4k_queue", f"3840x2160:dragon_standin_d9:default:shadow-queue:{G}"),
                 ("4k", f"3840x2160:dragon_standin_d9:default:{G}"), ("4k_compact", f"3840x2160:dragon_standin_d9:compact:{G}"),
                 ("1080p_display", f"1920x1080:dragon_standin_d9:display:{G}"), ("1080p_d10", f"1920x1080:dragon_standin_d10:default:{G}"),
                 ("cfg5", "3840x2160:dragon_standin_d9:spp16:grid8"),
                 ("1080p_dense", "1920x1080:dragon_standin_d9:default:grid9"), ("4k_dense", "3840x2160:dragon_standin_d9:default:grid9"),
                 ("1080p_split7", "1920x1080:dragon_standin_d9:default:grid7+blocks"), ("1080p_split6", "1920x1080:dragon_standin_d9:default:grid6+blocks"),
                 # scripts/profile_mesh.sh: the mesh-derived scenes (both get coarse level 7 + blocks)
                 ("mesh_d9", "1920x1080:knot_d9.asdf:default:grid7+blocks"), ("mesh_d9_4k", "3840x2160:knot_d9.asdf:default:grid7+blocks"),
                 ("mesh_d10", "1920x1080:knot_d10.asdf:default:grid7+blocks"), ("mesh_d10_4k", "3840x2160:knot_d10.asdf:default:grid7+blocks")]:
    import os
    if not os.path.isdir(f"gpurun_out/prof_{R}_{tag}"):
        continue
    subprocess.run([sys.executable, "scripts/summarise_profile.py", f"{R}_{tag}", key], stdout=subprocess.DEVNULL, check=False)
    print(tag, "done")
