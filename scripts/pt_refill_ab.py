"""Round 6, laboratory: cfg-5's bounce levels with LANE REFILL (k_pt_bounce_refill, SDFHIP_PT_REFILL=1) against the product's
k_pt_bounce: persistent waves whose lanes take the wave's next queue entry as soon as they have finished theirs, instead of one
entry per lane and the wave waiting for its slowest lane twice (lanes on in 44 % of the VALU thread-cycles).  Bit-identical
frames and identical counters expected; keep / drop rule: 5 % of the frame.

    python scripts/pt_refill_ab.py              A/B/A/B: ms per frame (3 frames in flight, as bench.py), frame and counters compared
    python scripts/pt_refill_ab.py one 0|1 N    N frames of one setting (for rocprofv3 passes around it)
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


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "ab"
    W, H = (3840, 2160) if os.environ.get("PT_AB_SIZE", "4k") == "4k" else (1920, 1080)
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
            os.environ["SDFHIP_PT_REFILL"] = sys.argv[2]
            frames(int(sys.argv[3]) if len(sys.argv) > 3 else 3)
            return
        ref = ref_counts = None
        for rnd, setting in enumerate(("0", "1", "0", "1")):
            os.environ["SDFHIP_PT_REFILL"] = setting
            frames(3)
            t0 = time.perf_counter()
            frames(9)
            ms = (time.perf_counter() - t0) / 9 * 1e3
            img = bufs[0].clone()
            st = sb.Stats()
            sc.DrawPathDevice(cam, W, H, bufs[1].data_ptr(), pt=pt, flags=sb.FLAG_COUNT, stream=streams[1].cuda_stream, stats=st)
            torch.cuda.synchronize()
            counts = (st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays, st.n_loads, st.n_hits)
            same_count_frame = bool(torch.equal(bufs[1].view(torch.int32), img.view(torch.int32)))
            if ref is None:
                ref, ref_counts = img, counts
            same = bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
            print(f"SDFHIP_PT_REFILL={setting}: {ms:8.3f} ms per {W}x{H} x 16 spp frame; identical to the product's frame: {same}; counting render "
                  f"identical: {same_count_frame}; counters equal: {counts == ref_counts}  {counts}", flush=True)


if __name__ == "__main__":
    main()
