#!/usr/bin/env python3
"""bench.py -- the ASDF sphere-tracing hot path on N MI355X (one process per GPU).

A "step" is one frame of the workload: every rank ray-marches its row bands of
the frame (scene and camera already resident in HBM), the band buffers are
gathered to rank 0 over RCCL and put back in row order.  At N=1 a step is the
ray-march kernel alone.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", default="1920x1080", help="frame WxH (BASELINE cfg-2: 1920x1080)")
    ap.add_argument("--depth", type=int, default=9, help="octree depth of the dragon stand-in")
    ap.add_argument("--asdf", default=None, help="render this .asdf instead of the synthetic scene")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "stack"])
    ap.add_argument("--compact", type=int, default=-1, help="wavefront ray compaction: 1 on, 0 off, -1 default")
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="consecutive frames are rendered on this many HIP streams (double buffering): the "
                         "long tail of one frame (a few 100-140-step pixels) overlaps the body of the next")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N>1 path through host buffers (not a perf mode)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify the assembled frame against a 1-GPU render")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import sdfbox_amd as sb
    from sdfbox_amd import _lib
    from sdfbox_amd.tiles import BandLayout, deinterleave, render_bands

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    ndev = torch.cuda.device_count()
    device = local_rank % ndev            # gloo rehearsal may put several ranks on one GPU
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    W, H = (int(v) for v in args.size.lower().split("x"))

    # ---- scene (host build, then resident in HBM before anything is timed) ----
    t0 = time.time()
    ncpu = os.cpu_count() or 1
    if args.asdf:
        od = sb.OctData.LoadAsdf(args.asdf)
        scene_name = os.path.basename(args.asdf)
    else:
        od = sb.dragon_standin(args.depth, nthreads=max(1, min(32, ncpu // max(1, min(world, ndev * 8)))))
        scene_name = f"dragon_standin_d{args.depth}"
    t_gen = time.time() - t0
    scene = sb.Scene(od, device=device)

    # ---- camera: SURVEY.md 8d cfg-2 --------------------------------------------
    cam = sb.Logic(W, H)
    cam.Position = (0.5, 0.5, -0.35)
    cam.Heading = (-0.2, 0.35)            # (X = pitch, Y = yaw), Logic.cs:53

    kflag = {"auto": sb.KERNEL_AUTO, "generic": sb.KERNEL_GENERIC, "stack": sb.KERNEL_STACK}[args.kernel]
    compact = (args.compact == 1) if args.compact >= 0 else DEFAULT_COMPACT
    flags = kflag | (sb.FLAG_COMPACT if compact else 0)

    layout = BandLayout(H, world, args.band_rows)
    # default: 2 frames in flight on one GPU; with the frame sharded over `world` GPUs each
    # rank's share is mostly tail, so keep more frames in flight
    nbuf = args.frames_in_flight if args.frames_in_flight > 0 else (2 if world == 1 else min(8, 2 * world))
    if world > 1:
        nbuf = max(2, nbuf)
    streams = [torch.cuda.Stream() for _ in range(nbuf)]     # one per frame in flight
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream
    local = [torch.zeros((layout.rows_per_rank if world > 1 else H, W, 4), dtype=torch.float32, device="cuda")
             for _ in range(nbuf)]
    gathered = frame = None
    if world > 1 and rank == 0:
        gathered = [torch.zeros((world, layout.rows_per_rank, W, 4), dtype=torch.float32, device="cuda")
                    for _ in range(nbuf)]
        frame = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(nbuf)]

    def render(buf, stats=None, fl=None, st=None):
        st = stream if st is None else st
        if world == 1:
            scene.DrawDevice(cam, W, H, buf.data_ptr(), flags=flags if fl is None else fl, stream=st, stats=stats)
        else:
            render_bands(scene, cam, W, layout, rank, buf.data_ptr(), flags=flags if fl is None else fl,
                         stream=st, stats=stats)

    pending = [None] * nbuf

    def finish(slot):
        """Complete the gather issued from buffer `slot` and, on rank 0, assemble the frame."""
        w = pending[slot]
        if w is None:
            return
        pending[slot] = None
        if args.backend == "nccl":
            with torch.cuda.stream(streams[slot]):
                w.wait()                              # this slot's stream waits for its gather
                if rank == 0:
                    deinterleave(device, gathered[slot].data_ptr(), frame[slot].data_ptr(), W, layout,
                                 stream=streams[slot].cuda_stream)
        else:                                         # gloo rehearsal: host buffers
            if rank == 0:
                g = torch.stack(w).cuda()
                gathered[slot].copy_(g)
                deinterleave(device, gathered[slot].data_ptr(), frame[slot].data_ptr(), W, layout, stream=stream)

    def step(k):
        slot = k % nbuf
        finish(slot)                                   # buffer reuse: its previous gather must be done
        render(local[slot], st=streams[slot].cuda_stream)
        if world > 1:
            if args.backend == "nccl":
                glist = list(gathered[slot].unbind(0)) if rank == 0 else None
                with torch.cuda.stream(streams[slot]):  # the collective orders itself after this stream's render
                    pending[slot] = dist.gather(local[slot], glist, dst=0, async_op=True)
            else:
                streams[slot].synchronize()
                host = local[slot].cpu()
                glist = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, glist, dst=0)
                pending[slot] = glist if rank == 0 else True
            if args.backend != "nccl":
                finish(slot)

    def drain():
        for s in range(nbuf):
            finish(s)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- algorithmic work of one frame (counting launch, untimed) ---------------
    st = sb.Stats()
    render(local[0], stats=st, fl=flags | sb.FLAG_COUNT)
    torch.cuda.synchronize()
    my_pixels = W * H if world == 1 else len(layout.rows_of(rank)) * W
    alg_bytes_rank = 8 * st.n_nodes + 8 * st.n_samples + 16 * my_pixels
    counters = torch.tensor([st.n_nodes, st.n_samples, st.n_steps, alg_bytes_rank], dtype=torch.float64)

    # ---- warm-up, then the timed region -------------------------------------------
    for k in range(args.warmup):
        step(k)
    drain()
    barrier()
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = counters.cuda() if args.backend == "nccl" else counters
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counters = c.cpu()

    # ---- per-launch kernel time, HIP events on the launch stream -----------------
    kms = []
    for _ in range(min(args.steps, 20)):
        render(local[0], stats=st)
        kms.append(st.kernel_ms)
    kernel_ms = float(np.mean(kms))
    torch.cuda.synchronize()

    check_ok = None
    if args.check and world > 1:
        if rank == 0:
            ref = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
            scene.DrawDevice(cam, W, H, ref.data_ptr(), flags=flags, stream=stream)
            torch.cuda.synchronize()
            check_ok = all(bool(torch.equal(f.view(torch.int32), ref.view(torch.int32))) for f in frame)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = W * H / (elapsed / args.steps) / 1e6
        peak = 8000.0                                  # GB/s, HBM3E spec (MI355X_MICROARCH.md)
        achieved = alg_bytes_rank / (kernel_ms * 1e-3) / 1e9
        traffic = load_traffic(W, H, scene_name, flags)
        out = {
            "metric": "Mray/s (primary rays; frame W*H / time per frame)",
            "value": round(mrays, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{W}x{H} primary-ray sphere trace + shadow march, {scene_name} "
                            f"(N={od.Length} nodes, {od.nbytes / 1e6:.1f} MB), camera (0.5,0.5,-0.35) yaw 0.35 pitch -0.2",
                "kernel": ("stack" if (st.kernel_used & 0xF) == sb.KERNEL_STACK else "generic") + ("+compact" if compact else ""),
                "parallelism": "1 GPU" if world == 1 else f"{world} GPUs, {args.band_rows}-row bands round-robin + gather to rank 0 ({args.backend})",
                "frames_in_flight": nbuf,
                "gstep_per_s": round(float(counters[2]) / (elapsed / args.steps) / 1e9, 3),
                "scene_build_s": round(t_gen, 2),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": peak,
                "unit": "GB/s",
                "frac": round(achieved / peak, 4),
                "traffic": traffic,
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": int(alg_bytes_rank),
            },
        }
        if check_ok is not None:
            out["config"]["assembled_frame_equals_1gpu"] = check_ok
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(od, cam, W, H, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# Compaction is the default only where it measured faster (DESIGN.md "Measurements").
DEFAULT_COMPACT = False


def load_traffic(W, H, scene_name, flags):
    """HBM bytes per launch from the committed PMC passes (profiles/), or None.
    bench.py cannot run rocprofv3 on itself; the passes are collected with
    scripts/profile_pmc.sh and summarised in profiles/hbm_traffic.json."""
    p = os.path.join(REPO, "profiles", "hbm_traffic.json")
    try:
        with open(p) as f:
            t = json.load(f)
        key = f"{W}x{H}:{scene_name}:{flags}"
        return t.get(key)
    except (OSError, ValueError):
        return None


def cpu_baseline(od, cam, W, H, target_seconds):
    """The CPU oracle (the build's C restatement of Compute.hlsl: the reference has
    no CPU path) timed on this host over a bounded sample of the
    same frame: every `step`-th row, rows interleaved over all hardware threads."""
    import oracle
    oracle.build()
    nthreads = min(os.cpu_count() or 1, 64)
    # calibrate on 2*nthreads rows spread over the frame, then size the sample
    ncal = min(H, 2 * nthreads)
    t0 = time.perf_counter()
    oracle.render(od.Structs, od.Values, cam.State, W, H, row0=0, nrows=ncal, row_step=max(1, H // ncal),
                  nthreads=nthreads)
    per_row = (time.perf_counter() - t0) / ncal
    rows = int(min(H, max(nthreads, target_seconds / max(per_row, 1e-9))))
    step = max(1, H // rows)
    nrows = (H + step - 1) // step
    t0 = time.perf_counter()
    oracle.render(od.Structs, od.Values, cam.State, W, H, row0=0, nrows=nrows, row_step=step, nthreads=nthreads)
    dt = time.perf_counter() - t0
    pix = nrows * W
    return {
        "value": round(pix / dt / 1e6, 3),
        "unit": "Mray/s",
        "cores": nthreads,
        "kind": "port",
        "sample": f"every {step}th row of the same {W}x{H} frame = {pix} pixels in {dt:.2f} s wall "
                  f"({dt * nthreads:.0f} core-seconds); oracle/sdf_oracle.c, gcc -O2 -ffp-contract=off, "
                  f"{nthreads} pthreads, rows interleaved",
    }


if __name__ == "__main__":
    main()
