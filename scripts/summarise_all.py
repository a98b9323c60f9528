"""python scripts/summarise_all.py r02  -- condense every gpurun_out/prof_<round>_* into profiles/ (see summarise_profile.py)."""
import subprocess, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
for tag, key in [("1080p", "1920x1080:dragon_standin_d9:default"), ("1080p_onekernel", "1920x1080:dragon_standin_d9:one-kernel"),
                 ("1080p_queue", "1920x1080:dragon_standin_d9:default:shadow-queue"), ("4k_queue", "3840x2160:dragon_standin_d9:default:shadow-queue"),
                 ("4k", "3840x2160:dragon_standin_d9:default"), ("4k_compact", "3840x2160:dragon_standin_d9:compact"),
                 ("1080p_display", "1920x1080:dragon_standin_d9:display"), ("1080p_d10", "1920x1080:dragon_standin_d10:default"),
                 ("cfg5", "3840x2160:dragon_standin_d9:spp16")]:
    subprocess.run([sys.executable, "scripts/summarise_profile.py", f"{R}_{tag}", key], stdout=subprocess.DEVNULL, check=False)
    print(tag, "done")
