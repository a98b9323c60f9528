"""Round 6's attempt at cfg-5's refetch factor (keep / drop rule: 5 % of the frame).

scripts/compulsory_bytes.py: a cfg-5 frame must move 9.7 GB (every distinct line of every launch once: 1.5 GB, + its queues and results: 8.1 GB)
and moves 120 GB; each bounce level touches 2.5-3.3 M lines (0.4 GB) chip-wide but 15-17 M summed over the XCDs -- every XCD
walks nearly the whole object -- and fetches each of them ~14 times.  The attempt: the bounce levels' entries in the order of
(region of the hit, octant of the outgoing direction) [round 4's key: SDFHIP_PT_SORT=R, no gain on its own], AND every XCD
walking a contiguous eighth of that order front to back (SDFHIP_PT_SORT_XCD=1, new): the workgroups that share an L2 then start
their rays in one part of the scene at a time.  Laboratory library; every setting is compared with the unsorted frame bit for bit.

    python scripts/pt_xcd_order_ab.py time            every setting: ms per frame (3 frames in flight, as bench.py), lines touched
    python scripts/pt_xcd_order_ab.py one R:X N       N frames of one setting (for a rocprofv3 --pmc pass around it)
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402
import sdfbox_amd.lab  # noqa: E402

sb = sdfbox_amd.lab.load()
W, H = 3840, 2160
SETTINGS = ["0:0", "3:0", "3:1", "2:1", "1:1"]


def setting(spec):
    r, x = spec.split(":")
    os.environ["SDFHIP_PT_SORT"] = r
    os.environ["SDFHIP_PT_SORT_XCD"] = x


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "time"
    od = sb.dragon_standin(9, nthreads=min(32, os.cpu_count() or 1))
    cam = bench.bench_camera(sb, W, H)
    pt = sb.PathTrace(spp=16)
    with sb.Scene(od, device=0) as sc:
        sb._lib.check(sb._lib.lib.sdfhip_scene_prepare_path(sc._h))
        nbuf = int(os.environ.get("PT_AB_FRAMES_IN_FLIGHT", "3"))
        streams = [torch.cuda.Stream() for _ in range(nbuf)]
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(nbuf)]

        def frames(n):
            for k in range(n):
                sc.DrawPathDevice(cam, W, H, bufs[k % nbuf].data_ptr(), pt=pt, stream=streams[k % nbuf].cuda_stream)
            torch.cuda.synchronize()
        if mode == "one":
            setting(sys.argv[2])
            frames(int(sys.argv[3]) if len(sys.argv) > 3 else 3)
            return
        ref = None
        for spec in SETTINGS:
            setting(spec)
            frames(3)
            t0 = time.perf_counter()
            frames(9)
            ms = (time.perf_counter() - t0) / 9 * 1e3
            img = bufs[0].clone()
            if ref is None:
                ref = img
            same = bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
            # the lines the counting pipeline touches under this order (chip-wide they cannot change; summed over the XCDs they can)
            sc.touch_begin()
            st = sb.Stats()
            sc.DrawPathDevice(cam, W, H, bufs[1].data_ptr(), pt=pt, flags=sb.FLAG_COUNT, stream=streams[1].cuda_stream, stats=st)
            torch.cuda.synchronize()
            t = sc.touch_end()
            lv = [(p["coarse_lines"] + p["fine_lines"], p["coarse_lines_xcd_sum"] + p["fine_lines_xcd_sum"]) for p in t["phases"][1:]]
            print(f"SDFHIP_PT_SORT={spec.split(':')[0]} SDFHIP_PT_SORT_XCD={spec.split(':')[1]}: {ms:8.3f} ms per frame, identical to unsorted: {same}; "
                  "bounce levels' lines chip-wide / summed over XCDs (M): " + ", ".join(f"{a / 1e6:.2f} / {b / 1e6:.2f}" for a, b in lv), flush=True)


if __name__ == "__main__":
    main()
