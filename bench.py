#!/usr/bin/env python3
"""bench.py -- the ASDF sphere-tracing hot path on N MI355X (one process per GPU).

A "step" is one frame of the workload: every rank ray-marches its row bands of
the frame (scene and camera already resident in HBM), the band buffers are
gathered to rank 0 over RCCL and put back in row order.  At N=1 a step is the
ray-march kernel alone.  Consecutive frames go to separate HIP streams round
robin (four frames in flight by default, on hardware queues of their own), as a renderer would keep them: the
long tail of one frame (a few 100-140-step pixels) overlaps the body of the
next ones.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # N > 1 without a launcher: spawns that command itself (see spawn_ranks)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# The driver's path is this file; what the line says about a measurement, the other configurations and the N > 1 machinery are
# modules beside it (their names stay importable from here: tests and scripts say bench.roofline, bench.orbit_cameras, ...).
from bench_report import (HBM_PEAK_GBS, KERNEL_SOURCES, VALU_PEAK_GINSTR, VALU_PEAK_SPEC_GINSTR, configs_summary,  # noqa: E402,F401
                          kernel_source_hash, load_compulsory, load_pmc, measured_hbm_bandwidth, roofline)
from bench_configs import bench_camera, grid_suffix, load_package, orbit_cameras, run_configs  # noqa: E402,F401
from bench_live_pmc import live_pmc, merged as merge_pmc  # noqa: E402
from bench_sustained import GpuTelemetry, at_observed_clock, sustained_leg, sustained_seconds  # noqa: E402,F401
from bench_sharded import Watchdog, link_check, main_single_process, measure_band_deal, measure_rank0_weight, spawn_ranks  # noqa: E402,F401

# Wavefront ray compaction is measured 2-3 % slower than the plain kernel on this workload
# (DESIGN.md section 4.4), so it is off unless asked for.
DEFAULT_COMPACT = False


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--size", default="1920x1080", help="frame WxH (BASELINE cfg-2: 1920x1080)")
    ap.add_argument("--depth", type=int, default=9, help="octree depth of the dragon stand-in")
    ap.add_argument("--asdf", default=None, help="render this .asdf instead of the synthetic scene")
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "stack"])
    ap.add_argument("--top-grid-level", type=int, default=None,
                    help="A/B: upload the scene with a plain lookup grid of this level (sdfhip_upload_options; the tree's depth = the dense full-depth grid)")
    ap.add_argument("--top-grid-split", type=int, default=None, help="A/B: upload the scene with a split grid of this coarse level")
    ap.add_argument("--compact", type=int, default=-1, help="wavefront ray compaction: 1 on, 0 off, -1 default")
    ap.add_argument("--band-rows", type=int, default=16)
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="HIP streams / buffers used round-robin (0 = default: 4; 2 for --spp / --compact).  Unsharded: frames in flight.  "
                         "Sharded: groups of --gather-every frames in flight")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="sharded runs: frames per gather (fewer, larger messages; one collective launch per "
                         "group instead of per frame).  0 = default: 4")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearse the N>1 path through host buffers (not a perf mode)")
    ap.add_argument("--exercise-gather", action="store_true",
                    help="run the sharded path (bands + gather + de-interleave) even with one rank")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall-time target of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--configs", default="auto", choices=["auto", "all", "none"],
                    help="the other single-GPU BASELINE configurations timed in the same process, behind the headline: cfg-3 (4K, "
                         "compaction off and on), cfg-5 (4K, 16 spp), cfg-2 on the depth-10 scene -> the line's `configs` block.  "
                         "auto: when the command is the headline's (1 GPU, 1920x1080, default scene and kernel); all: always; none")
    ap.add_argument("--configs-scale", type=int, default=1,
                    help="tests only: the `configs` block on frames 1/N the size and the deeper scene at --depth + 1 (no PMC pass matches)")
    ap.add_argument("--display", action="store_true",
                    help="fuse SdfBox's display pass (DisplayFrag.hlsl) into the epilogue: RGBA8 frames, 4x fewer "
                         "bytes stored and gathered (SURVEY 8f N2); the headline metric is measured without it")
    ap.add_argument("--spp", type=int, default=0,
                    help="path-traced mode (BASELINE config 5: --size 3840x2160 --spp 16): spp jittered camera "
                         "rays per pixel + 3 diffuse bounces; Mray/s then counts W*H*spp camera rays")
    ap.add_argument("--orbit", type=int, default=0,
                    help="timed region with a camera that moves every frame: N precomputed Info blocks on a circle "
                         "around the scene's centre (1 degree apart), used round-robin (0 = the fixed cfg-2 camera; "
                         "the default line reports an orbit pass beside it in `latency`)")
    ap.add_argument("--tile-order", action="store_true",
                    help="render every frame with SDFHIP_FLAG_TILE_ORDER (tiles in descending order of their cost in the stream's "
                         "last frame); without it only the latency.tile_order figures use the flag")
    ap.add_argument("--no-tile-order", action="store_true",
                    help="A/B: sharded runs (N > 1) launch their shares' tiles in the default order instead of by last launch's cost")
    ap.add_argument("--shadow-queue", action="store_true",
                    help="A/B knob: k_march queues its shadow rays for a second kernel (k_shadow), 64 to a wave")
    ap.add_argument("--one-kernel", action="store_true",
                    help="A/B: round 1's one-kernel form (lane state machine, shading in place) instead of the default "
                         "k_march (wave-converged march / shading / shadow march)")
    ap.add_argument("--check", action="store_true", help="verify the assembled frame against a whole-frame render (always done when several ranks take part)")
    ap.add_argument("--rank0-weight", type=float, default=0.0,
                    help="sharded runs: rank 0's share of the frame as a fraction of a peer's share "
                         "(0 = measure at start-up so that render + assembly on rank 0 takes as long as a peer's render)")
    ap.add_argument("--deal", default="cost", choices=["cost", "weight"],
                    help="sharded runs: how the frame's bands are dealt -- 'cost': by their measured cost, longest first, rank 0 charged for "
                         "the assembly (tiles.balanced_owner); 'weight': round robin by credit with rank 0's share a searched fraction of a "
                         "peer's (rounds 2-4; also what the other output modes use)")
    ap.add_argument("--sparse-cap-scale", type=float, default=1.25,
                    help="sparse shares: the floats that travel with a share = this x the lit pixels measured before the timed region (a value "
                         "below 1 forces tails to be sent again)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 (or --devices) through the library's own multi-device entry points (sdfhip_multi_*: one process, one "
                         "host thread and stream per device, peer copies or RCCL inside the library) instead of one process per GPU "
                         "under torch.distributed; the A/B of the two host designs")
    ap.add_argument("--devices", default=None,
                    help="--single-process: the device list, e.g. 0,0,0,0 to rehearse four ranks on one GPU (default 0..gpus-1)")
    ap.add_argument("--multi-mode", default="groups", choices=["groups", "frame"],
                    help="--single-process: 'groups' = groups of --gather-every frames, four groups in flight (throughput); "
                         "'frame' = one frame at a time across all devices, launch to completion (what a viewer waits for)")
    ap.add_argument("--init-timeout", type=float, default=90.0,
                    help="N > 1: seconds the rendezvous (init_process_group) and every later collective may take before it raises")
    ap.add_argument("--link-timeout", type=float, default=30.0,
                    help="N > 1: seconds a peer's first 1 MB message may take to reach rank 0 in the link check before every rank gives up "
                         "(exit code 5, the link named).  Generous on purpose: the first message between two ranks also sets their channel up "
                         "(hundreds of milliseconds on one GPU; unknown on eight), and a false alarm would cost the run")
    ap.add_argument("--watchdog-seconds", type=float, default=-1.0,
                    help="a daemon thread writes the run's phase to stderr every few seconds and ends the process with exit code 3 once "
                         "the whole run has taken this long (-1 = default: 420 for N > 1, off for N = 1; 0 = off)")
    ap.add_argument("--only-timed", action="store_true",
                    help="skip the passes that render OTHER frames than the timed region's before it (the moving-camera and tile-order latency "
                         "figures): for a profiler pass, whose per-frame counters are averages over every launch of the process "
                         "(scripts/profile.sh passes it; until round 6 its averages included those passes' frames)")
    ap.add_argument("--sustained", default="auto",
                    help="continuous-operation legs behind the headline (bench_sustained.py): 'cfg2,cfg3,orbit' seconds of wall time, e.g. "
                         "'5,3,2' -- the cfg-2 frame pipelined for >= 5 s, the 4K frame (in the `configs` block) for >= 3 s, the cfg-2 frame "
                         "with a camera that moves every frame for >= 2 s, the GPU's clock / power / temperature sampled every 100 ms.  "
                         "auto: 5,3,2 when the command is the headline's, else off; 0 or off: none")
    ap.add_argument("--live-pmc", default="auto", choices=["auto", "on", "all", "off"],
                    help="the roofline's counters measured in this run (bench_live_pmc.py): after the timed region, the headline's command is run "
                         "again as child processes under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE, the VALU counters: one pass each, 5 frames) and "
                         "`roofline.traffic` comes from those passes, the committed figure beside it.  on: the headline's command only (about 8 s "
                         "more); auto / all: when the command is the headline's and rocprofv3 is there, also the moving-camera pass and every entry "
                         "of the `configs` block, each through child passes of its own command line (about a minute more); off: the committed "
                         "passes only (--only-timed implies off)")
    ap.add_argument("--lab", action="store_true",
                    help="load the experiments flavour of the library (libsdfhip_lab.so, include/sdfhip_experimental.h): needed by the A/B "
                         "forms --one-kernel and --shadow-queue")
    return ap.parse_args()


def main():
    args = parse()
    # Exactly one line may reach stdout: the JSON.  Libraries loaded below print banners there
    # (RCCL's version block, for one), so fd 1 points at stderr until the line is written.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_process or args.devices:
        if "WORLD_SIZE" in os.environ and world > 1:
            raise SystemExit("--single-process is one process for all devices: start it without a launcher")
        return main_single_process(args, json_fd)
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            os.dup2(json_fd, 1)
            raise SystemExit(spawn_ranks(args.gpus))   # nothing has touched the GPU yet in this process
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    budget = args.watchdog_seconds if args.watchdog_seconds >= 0 else (420.0 if world > 1 else 0.0)
    wd = Watchdog(rank, budget)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver supports dmabuf IPC only (RCCL needs it)
    # HIP deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); streams that share a queue run their
    # kernels one behind the other.  A renderer that keeps frames in flight wants them on queues of their own: with 8 queues, four
    # frames in flight give 0.0871 ms per 1080p frame where six streams on four queues gave 0.0881 (and three streams 0.1136: two of
    # them on one queue) -- scripts/sweep_driver_style.sh, DESIGN.md section 6.  Read by the runtime when it starts: set before torch loads it.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    sb = load_package(args)
    BandLayout, deinterleave, deinterleave_sparse2, render_bands, render_bands_batch, render_sparse2, sparse2_bytes, sparse2_floats_offset = (
        sb.tiles.BandLayout, sb.tiles.deinterleave, sb.tiles.deinterleave_sparse2, sb.tiles.render_bands, sb.tiles.render_bands_batch,
        sb.tiles.render_sparse2, sb.tiles.sparse2_bytes, sb.tiles.sparse2_floats_offset)
    SparseShareCall, SparseExpandCall = sb.tiles.SparseShareCall, sb.tiles.SparseExpandCall

    import datetime
    sharded = world > 1 or args.exercise_gather
    nccl = args.backend == "nccl"

    def rendezvous(**kw):
        """first contact: every rank must arrive within --init-timeout, or this rank says so and ends non-zero"""
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        wd.phase(f"rendezvous ({args.backend} init_process_group, world {world}, {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')})",
                 limit=args.init_timeout + 30.0)
        try:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=args.init_timeout), **kw)
        except Exception as e:
            print(f"[bench rank {rank}] rendezvous failed after at most {args.init_timeout:.0f} s: not every one of the {world} ranks "
                  f"arrived ({type(e).__name__}: {str(e)[:300]})", file=sys.stderr, flush=True)
            raise SystemExit(4)
    links = None
    if sharded and not nccl:
        rendezvous()                      # gloo needs no GPU: before anything touches one
        if world > 1:                     # ... nor does the rehearsal of the link check (bench_sharded.link_check): through host buffers
            links = link_check(torch, dist, rank, world, False, wd, ["host buffers"] * world, limit=args.link_timeout)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    ndev = torch.cuda.device_count()
    device = local_rank % ndev            # the gloo rehearsal may put several ranks on one GPU
    torch.cuda.set_device(device)
    if sharded and nccl:
        rendezvous(device_id=torch.device("cuda", device))

    W, H = (int(v) for v in args.size.lower().split("x"))
    # who really takes part (the line says so: a scaling curve is only one if the ranks sit on different GPUs)
    wd.phase("who takes part (all_gather of the PCI bus ids: the first collective)", limit=args.init_timeout + 30.0)
    bus_ids = [sb.device_pci_bus_id(device)]
    if sharded and world > 1:
        got = [None] * world
        dist.all_gather_object(got, bus_ids[0])
        bus_ids = got
        if nccl:
            # first contact with the links themselves, before the weight search times anything over them: 1 MB from every peer into
            # rank 0, every byte compared, 10 s per link, a per-link table on stderr
            peer = [None] * world
            try:
                mine = bool(torch.cuda.can_device_access_peer(device, 0 % ndev)) if device != 0 % ndev else True
            except Exception:
                mine = None
            dist.all_gather_object(peer, mine)
            links = link_check(torch, dist, rank, world, True, wd, bus_ids, limit=args.link_timeout, peer_access=peer)

    wd.phase("scene build + upload", quiet=True)
    # ---- scene: built on the host, resident in HBM before anything is timed ----------
    t0 = time.time()
    ncpu = os.cpu_count() or 1
    if args.asdf:
        od = sb.OctData.LoadAsdf(args.asdf)
        scene_name = os.path.basename(args.asdf)
    else:
        od = sb.dragon_standin(args.depth, nthreads=max(1, min(32, ncpu // max(1, min(world, ndev * 8)))))
        scene_name = f"dragon_standin_d{args.depth}"
    t_gen = time.time() - t0
    scene = sb.Scene(od, device=device, top_grid_level=args.top_grid_level, top_grid_split=args.top_grid_split)

    # ---- camera: SURVEY.md 8d cfg-2 ---------------------------------------------------
    cam = sb.Logic(W, H)
    cam.Position = (0.5, 0.5, -0.35)
    cam.Heading = (-0.2, 0.35)            # (X = pitch, Y = yaw), Logic.cs:53
    # --orbit N: the camera of frame k is orbit[k % N] -- the cfg-2 camera carried round the scene's centre in
    # steps of one degree (position on the circle through (0.5, 0.5, -0.35) about the vertical axis, yaw turned
    # by the same angle, so the object stays in view as it does when a user walks round it, Logic.cs:252-271)
    n_orbit_default = 90
    orbit = orbit_cameras(sb, W, H, args.orbit if args.orbit > 0 else n_orbit_default)
    cams = orbit if args.orbit > 0 else [cam]

    def cam_of(k):
        return cams[k % len(cams)]

    kflag = {"auto": sb.KERNEL_AUTO, "generic": sb.KERNEL_GENERIC, "stack": sb.KERNEL_STACK}[args.kernel]
    compact = (args.compact == 1) if args.compact >= 0 else DEFAULT_COMPACT
    flags = kflag | (sb.FLAG_COMPACT if compact else 0) | (sb.FLAG_DISPLAY if args.display else 0) | \
        (sb.TUNE_ONE_KERNEL if args.one_kernel else 0) | (sb._lib.TUNE_SHADOW_QUEUE if args.shadow_queue else 0) | \
        (sb.FLAG_TILE_ORDER if args.tile_order else 0)
    px_shape, px_dtype, px_bytes = ((), torch.int32, 4) if args.display else ((4,), torch.float32, 16)
    pt = sb.PathTrace(spp=args.spp) if args.spp > 0 else None
    # what travels in the gather: the sparse shares the march kernel writes itself (sdfhip_render_sparse_device) wherever the scene
    # has a grid as deep as the tree and the frame is the default kernel's; else the frame's own pixels (path-traced mode, display
    # pass, compaction, trees without such a grid)
    sparse2 = sharded and not args.display and pt is None and not compact and not args.one_kernel and not args.shadow_queue and \
        scene.top_grid_level > 0 and scene.stack_kernel_ok

    def share_shape(rows):                 # one dense frame-share of `rows` rows as it is rendered and gathered
        return (rows, W) + px_shape
    if pt is not None and (args.display or compact):
        raise SystemExit("--spp excludes --display and --compact")

    # sharded: G frames share one launch (grid.y = frame) and one gather; nbuf groups are in
    # flight (so G*nbuf frames).  A rank's share is mostly the serial tail of its longest
    # pixels: G frames in one grid share that tail
    # (8 at 8 ranks: the shares are small there, and a group costs ~60 us of host time: launch, collective, wait)
    plan_G, plan_nbuf, plan_ordered = sb.tiles.group_plan(world, args.steps)
    G = (args.gather_every if args.gather_every > 0 else plan_G) if sharded else 1
    if pt is not None or compact:
        G = 1 if not sharded else G            # those kernels render one frame per launch
    # in flight: 4 frames on one GPU, on hardware queues of their own (GPU_MAX_HW_QUEUES above): the long tail of a frame's last waves
    # runs under the body of the next ones -- 1080p 0.163 ms with one frame in flight, 0.0896 / 0.0872 / 0.0871 with two / three / four,
    # 0.0998 with five (8 queues; profiles/r04_hw_queues.txt in the history, commit 53ee955.  On the runtime's default of 4 queues: 0.0898 with four, 0.0881 with
    # six, 0.1136 with three); 3 for the path-traced mode (its buffers are gigabytes per stream; 21.6 / 20.9 / 20.9 / 20.8 ms per cfg-5 frame
    # with 2 / 3 / 4 / 5 -- on four hardware queues the third brought nothing); 4
    # groups of a sharded run (scripts/batch_sweep.py: a rank's share of 4 frames per launch needs 4 launches in flight to fill the
    # chip -- 0.039 -> 0.024 ms per frame at 8 ranks)
    nbuf = args.frames_in_flight if args.frames_in_flight > 0 else (plan_nbuf if sharded else 3 if pt is not None else 4)
    # the shares of a SHORT sharded run launch their tiles in the order of their cost in the stream's last launch (SDFHIP_FLAG_TILE_ORDER
    # on the batched launch): a short run ends with its longest waves -- they start first (tiles.group_plan says when)
    if sparse2 and world > 1 and not args.no_tile_order and (plan_ordered or args.tile_order):
        flags |= sb.FLAG_TILE_ORDER

    # rank 0 also assembles the frame (de-interleave + wire expansion of every rank's rows), so it
    # renders a smaller share: --rank0-weight, or measured here before anything is timed
    wd.phase("rank-0 weight search (rank 0 times shares; the others wait in a broadcast)")
    w0 = args.rank0_weight if world > 1 else 1.0
    deal = None
    n_bands_frame = (H + args.band_rows - 1) // args.band_rows
    if sharded and world > 1 and w0 <= 0:
        if sparse2 and args.deal == "cost" and world <= n_bands_frame <= 512:
            # the bands dealt by their measured cost, rank 0 charged for the assembly (tiles.balanced_owner)
            deal = measure_band_deal(sb, scene, cam, W, H, world, args.band_rows, flags, rank, nccl, G, nbuf, args.steps, burst=plan_ordered)
        else:
            w0 = measure_rank0_weight(sb, scene, cam, W, H, world, args.band_rows, flags, share_shape, px_dtype, px_bytes,
                                      rank, nccl, pt, compact, G, nbuf, sparse2)
    if n_bands_frame > 512 or w0 > 0.98 or w0 <= 0:
        w0 = 1.0
    wd.phase("buffers, first share, channel set-up (the first send / recv between two ranks)")
    layout = BandLayout(H, world, args.band_rows, w0, owner=deal)
    streams = [torch.cuda.Stream() for _ in range(nbuf)]                       # one per group in flight
    main = torch.cuda.current_stream().cuda_stream
    rows_local = layout.rows_per_rank if sharded else H
    local = [torch.zeros((G,) + share_shape(rows_local), dtype=px_dtype, device="cuda") for _ in range(nbuf)] if not sparse2 else None
    gathered = frame = None
    # sparse2: a share holds the G frames of a group and room for every float; what travels is its fixed part and the first
    # `s2_send` floats (measured now, x --sparse-cap-scale); a share that needed more sends the tail again, point to point
    s2 = None
    if sparse2:
        full = rows_local * W * G
        s2 = {"full": full, "bytes": sparse2_bytes(rows_local, W, G, full), "off": sparse2_floats_offset(rows_local, W, G),
              "share": [torch.zeros(sparse2_bytes(rows_local, W, G, full), dtype=torch.uint8, device="cuda") for _ in range(nbuf)],
              "base": [0] * nbuf, "own": [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(nbuf)],
              "ev": [torch.cuda.Event() for _ in range(nbuf)],
              # a partial last group (steps not a multiple of G) renders and expands its own frames only: the share's layout is
              # that of n frames then (its floats start earlier in the buffer), the bytes that travel stay the same
              "n": [G] * nbuf, "off_of": {n: sparse2_floats_offset(rows_local, W, n) for n in range(1, G + 1)}}
        render_sparse2(scene, [cam] * G, W, layout, rank, s2["share"][0].data_ptr(), full, 0, flags=flags, stream=main)
        torch.cuda.synchronize()
        used = int(s2["share"][0][:4].view(torch.int32).item())
        s2["share"][0][:4].zero_()             # (every rank's counters start at zero: rank 0 keeps the peers' bases from there)
        torch.cuda.synchronize()
        need = torch.tensor([used], dtype=torch.int64, device="cuda" if nccl else "cpu")
        if world > 1:
            dist.all_reduce(need, op=dist.ReduceOp.MAX)
        s2["send"] = min(full, max(1024, (int(int(need.item()) * args.sparse_cap_scale) + 1023) // 1024 * 1024))
        s2["prefix"] = s2["off"] + 4 * s2["send"]
        if rank == 0:
            s2["gath"] = [torch.zeros((world, s2["bytes"]), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
            s2["counts"] = [torch.zeros(world, dtype=torch.int32).pin_memory() for _ in range(nbuf)]
            s2["bases"] = [[0] * world for _ in range(nbuf)]
    if sharded and world > 1 and sparse2:
        # the point-to-point path of a resend, used once before anything is timed: the first send / recv between two ranks sets
        # their channel up (with NCCL: hundreds of milliseconds), and the first overflow of a run may come inside the timed region
        probe = torch.zeros(1, dtype=torch.int32, device="cuda" if nccl else "cpu")
        if rank == 0:
            for r in range(1, world):
                dist.recv(probe, src=r)
        else:
            dist.send(probe, dst=0)
        if nccl:
            torch.cuda.synchronize()
    resent = 0
    if sharded and rank == 0:
        if not sparse2:
            gathered = [torch.zeros((world,) + tuple(local[0].shape), dtype=local[0].dtype, device="cuda") for _ in range(nbuf)]
        frame = [torch.zeros((G, H, W) + px_shape, dtype=px_dtype, device="cuda") for _ in range(nbuf)]

    def s2_ptrs(slot):                     # rank 0 reads its own share where it rendered it
        return [s2["share"][slot].data_ptr()] + [s2["gath"][slot][r].data_ptr() for r in range(1, world)]
    if sparse2:
        # what a group's launch, gather and expansion need, made once: the host's work per group is a handful of calls, none of
        # which builds an argument array or a tensor view (a rank's share of a 1080p group at 8 ranks is ~90 us of GPU work)
        s2["call"] = SparseShareCall(scene, W, layout, rank, s2["full"], max_frames=G, flags=flags)
        s2["mine"] = [s2["share"][sl][:s2["prefix"]] for sl in range(nbuf)]
        s2["own_src"] = [s2["share"][sl][:4].view(torch.int32) for sl in range(nbuf)]
        s2["ptr"] = [s2["share"][sl].data_ptr() for sl in range(nbuf)]
        if rank == 0:
            s2["expand"] = SparseExpandCall(device, W, layout, s2["full"])
            s2["ptrs"] = [s2_ptrs(sl) for sl in range(nbuf)]
            s2["glist"] = [[s2["gath"][sl][r][:s2["prefix"]] for r in range(world)] for sl in range(nbuf)]
            s2["frame_ptr"] = [frame[sl].data_ptr() for sl in range(nbuf)]
            s2["counts_ptr"] = [s2["counts"][sl].data_ptr() for sl in range(nbuf)]
    static_groups = {}

    def group_of(k, n):                    # the cameras of the n frames that end at step k; a camera at rest: one list object per n
        if len(cams) == 1:
            key = (id(cams[0]), n)
            if key not in static_groups:
                static_groups[key] = [cams[0]] * n
            return static_groups[key]
        return [cam_of(k - n + 1 + i) for i in range(n)]

    def assemble(slot, st):
        deinterleave(device, gathered[slot].data_ptr(), frame[slot].data_ptr(), W, layout, stream=st, pixel_bytes=px_bytes, frames=G)

    def render(buf, st, stats=None, fl=None, c=None):
        f = flags if fl is None else fl
        c = cam if c is None else c
        if sharded:
            render_bands(scene, c, W, layout, rank, buf.data_ptr(), flags=f, stream=st, stats=stats, pt=pt)
        elif pt is not None:
            scene.DrawPathDevice(c, W, H, buf.data_ptr(), pt=pt, flags=f, stream=st, stats=stats)
        else:
            scene.DrawDevice(c, W, H, buf.data_ptr(), flags=f, stream=st, stats=stats)

    pending = [None] * nbuf
    asm_ev = []                           # rank 0: (start, end, frames) HIP events around each assembly (expansion of all shares / de-interleave)
    ev = []                               # (start, end) HIP events around each timed launch
    rendered = 0                          # frames the timed launches rendered (= steps: a partial last group renders its own frames only)

    batched = sharded and pt is None and not compact     # one launch per group (else one per frame)

    host_t = {"launch": 0.0, "gather": 0.0, "finish": 0.0, "groups": 0}     # host seconds in the timed region's calls, by part

    def finish(slot):
        """Complete the gather issued from group buffer `slot`; rank 0 puts the rows of its G frames
        back in order."""
        w, pending[slot] = pending[slot], None
        if w is None:
            return
        st = streams[slot]
        if sparse2:
            t_f = time.perf_counter()
            finish_sparse2(slot, w, st)
            host_t["finish"] += time.perf_counter() - t_f
            return
        if nccl:
            with torch.cuda.stream(st):
                w.wait()                              # the group's stream waits for its gather
                if rank == 0:
                    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a0.record(st)
                    assemble(slot, st.cuda_stream)
                    a1.record(st)
                    asm_ev.append((a0, a1, G))
        elif rank == 0:                               # gloo rehearsal: through host buffers
            gathered[slot].copy_(torch.stack(w).cuda())
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            assemble(slot, main)
            a1.record()
            asm_ev.append((a0, a1, G))

    def finish_sparse2(slot, w, st):
        """the gather of sparse2 shares: rank 0 expands them; a share whose floats did not all travel sends the tail again"""
        nonlocal resent
        n = s2["n"][slot]                                   # frames of this slot's group
        off = s2["off_of"][n]
        nsend = (s2["prefix"] - off) // 4                   # floats that travelled: the same bytes, behind a shorter fixed part
        if nccl:
            with torch.cuda.stream(st):
                w.wait()
                if rank == 0:
                    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a0.record(st)
                    s2["expand"](slot, s2["ptrs"][slot], s2["frame_ptr"][slot], frames=n, counts_ptr=s2["counts_ptr"][slot], stream=st.cuda_stream)
                    a1.record(st)
                    asm_ev.append((a0, a1, n))
                    s2["ev"][slot].record(st)
        elif rank == 0:                                   # gloo rehearsal: through host buffers
            for r in range(1, world):
                s2["gath"][slot][r][:s2["prefix"]].copy_(w[r].cuda())
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            deinterleave_sparse2(device, s2_ptrs(slot), frame[slot].data_ptr(), W, layout, s2["full"], frames=n,
                                 counts_ptr=s2["counts"][slot].data_ptr(), stream=main)
            a1.record()
            asm_ev.append((a0, a1, n))
            torch.cuda.synchronize()
        if rank == 0:
            if nccl:
                s2["ev"][slot].synchronize()
            counters = [int(c) & 0xFFFFFFFF for c in s2["counts"][slot].tolist()]
            for r in range(world):
                used = (counters[r] - s2["bases"][slot][r]) & 0xFFFFFFFF
                s2["bases"][slot][r] = counters[r]
                if r == 0:
                    s2["base"][slot] = counters[0]
                    continue
                if used > nsend:                          # the tail of rank r's floats, then its rows again
                    resent += 1
                    tail = s2["gath"][slot][r][off + 4 * nsend: off + 4 * min(used, s2["full"])]
                    if nccl:
                        with torch.cuda.stream(st):
                            dist.recv(tail, src=r)
                    else:
                        host = torch.empty(tail.shape, dtype=tail.dtype)
                        dist.recv(host, src=r)
                        tail.copy_(host)
                    with torch.cuda.stream(st):
                        deinterleave_sparse2(device, s2_ptrs(slot), frame[slot].data_ptr(), W, layout, s2["full"], frames=n, only_rank=r,
                                             stream=(st.cuda_stream if nccl else main))
        else:
            s2["ev"][slot].synchronize()
            counter = int(s2["own"][slot].item()) & 0xFFFFFFFF
            used = (counter - s2["base"][slot]) & 0xFFFFFFFF
            s2["base"][slot] = counter
            if used > nsend:
                tail = s2["share"][slot][off + 4 * nsend: off + 4 * min(used, s2["full"])]
                if nccl:
                    with torch.cuda.stream(st):
                        dist.send(tail, dst=0)
                else:
                    dist.send(tail.cpu(), dst=0)

    def step(k, timed=False, last=False):
        nonlocal rendered
        group, within = divmod(k, G)
        slot = group % nbuf
        if within == 0:
            finish(slot)                              # the slot's previous group must be complete
        s = streams[slot]
        group_ends = within == G - 1 or last
        if batched:
            if not group_ends:
                return                                # the whole group is one launch, issued at its end
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
            group = group_of(k, within + 1)
            if sparse2:
                s2["n"][slot] = len(group)
                t_l = time.perf_counter()
                s2["call"](group, s2["ptr"][slot], s2["base"][slot], stream=s.cuda_stream, flags=flags)
                host_t["launch"] += time.perf_counter() - t_l
                host_t["groups"] += 1
            else:
                render_bands_batch(scene, group, W, layout, rank, local[slot].data_ptr(), flags=flags, stream=s.cuda_stream)
            if timed:
                e1.record(s)
                ev.append((e0, e1, within + 1))
                rendered += len(group)
        else:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
            render(local[slot][within], s.cuda_stream, c=cam_of(k))
            if timed:
                e1.record(s)
                ev.append((e0, e1, 1))
                rendered += 1
        if not sharded or not group_ends:
            return
        if sparse2:
            t_g = time.perf_counter()
            if rank != 0:                             # this rank's own counter, for finish()
                with torch.cuda.stream(s):
                    s2["own"][slot].copy_(s2["own_src"][slot], non_blocking=True)
                    s2["ev"][slot].record(s)
            mine = s2["mine"][slot]
            if nccl:
                glist = s2["glist"][slot] if rank == 0 else None
                with torch.cuda.stream(s):
                    pending[slot] = dist.gather(mine, glist, dst=0, async_op=True)
            else:
                s.synchronize()
                if rank != 0:
                    s2["ev"][slot].synchronize()
                host = mine.cpu()
                glist = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, glist, dst=0)
                pending[slot] = glist if rank == 0 else True
                finish(slot)
            host_t["gather"] += time.perf_counter() - t_g
            return
        # one collective for the whole group (a partial last group is gathered whole, too)
        if nccl:
            glist = list(gathered[slot].unbind(0)) if rank == 0 else None
            with torch.cuda.stream(s):                # the collective orders itself behind this stream's renders
                pending[slot] = dist.gather(local[slot], glist, dst=0, async_op=True)
        else:
            s.synchronize()
            host = local[slot].cpu()
            glist = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, glist, dst=0)
            pending[slot] = glist if rank == 0 else True
            finish(slot)

    def drain():
        for s in range(nbuf):
            finish(s)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wd.phase("counting render", quiet=True)
    # ---- algorithmic work of one frame: counting build of the same kernel, untimed ------
    st = sb.Stats()
    count_buf = local[0] if not sparse2 else torch.zeros((rows_local, W, 4), dtype=torch.float32, device="cuda")
    render(count_buf, main, stats=st, fl=flags | sb.FLAG_COUNT)
    torch.cuda.synchronize()
    del count_buf
    my_pixels = len(layout.rows_of(rank)) * W if sharded else W * H
    # the bytes the REFERENCE algorithm reads and writes for these pixels (Compute.hlsl:88-108: entry +
    # ascents + descents, 8 B of topology each; 8 B of values per sample; the pixel store)
    ref_bytes_rank = 8 * st.n_nodes + 8 * st.n_samples + px_bytes * my_pixels
    # the bytes THIS kernel's own algorithm moves: 16 B per grid cell / node record a lane loads, the pixel store, and
    # under --shadow-queue 64 B written and 64 B read per shadow ray queued between the two kernels
    own_bytes_rank = 16 * st.n_loads + 128 * st.n_hits + px_bytes * my_pixels
    counters = torch.tensor([st.n_nodes, st.n_samples, st.n_steps, st.n_shadow_rays, st.n_loads, st.n_hits], dtype=torch.float64)
    kernel_used = st.kernel_used

    # ---- latency: one frame (sharded: one gather group) at a time, nothing else in flight, host clock around
    # launch + completion; and the same with the camera moving every frame.  Measured BEFORE the timed region: a few
    # hundred frames that also bring the GPU's clocks up, so that a short timed run (--steps 20 --warmup 5 is 2.5 ms of
    # work) measures the steady state the long ones do ------------------------------------------------------------
    def latency_pass(camlist, n, extra_flags=0):
        nonlocal cams, flags
        saved, cams = cams, camlist
        saved_flags, flags = flags, flags | extra_flags
        times = []
        try:
            for j in range(n + 3):
                barrier()
                t0 = time.perf_counter()
                for w in range(G):
                    step(j * G + w, last=(w == G - 1))
                drain()
                barrier()
                if j >= 3:
                    times.append(time.perf_counter() - t0)
        finally:
            cams = saved
            flags = saved_flags
        t = torch.tensor([float(np.median(times))], dtype=torch.float64, device="cuda" if (nccl and world > 1) else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3

    def throughput_pass(camlist, n):
        nonlocal cams
        saved, cams = cams, camlist
        try:
            for k in range(min(n, 2 * G * nbuf)):
                step(k, last=(k == min(n, 2 * G * nbuf) - 1))
            drain()
            barrier()
            t0 = time.perf_counter()
            for k in range(n):
                step(k, last=(k == n - 1))
            drain()
            barrier()
            dt = time.perf_counter() - t0
        finally:
            cams = saved
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if (nccl and world > 1) else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / n * 1e3

    wd.phase("latency passes (one group at a time)")
    n_lat = 5 if pt is not None else 40
    latency = {"frames": G, "ms": round(latency_pass(cams, n_lat), 4)}
    if pt is None and not args.check and not args.only_timed:
        latency["orbit_cameras"] = len(orbit)
        latency["orbit_ms"] = round(latency_pass(orbit, n_lat), 4)
        latency["orbit_ms_per_step"] = round(throughput_pass(orbit, max(args.steps, 2 * len(orbit))), 4)
        if not sharded and not compact and not args.one_kernel:
            # SDFHIP_FLAG_TILE_ORDER (tiles in descending order of their cost in the stream's last frame): the same passes with it,
            # and both ways on a slower orbit -- a quarter of a degree per frame is still four turns per second at 0.17 ms per frame
            slow = orbit_cameras(sb, W, H, 120, 0.25)
            latency["tile_order"] = {"ms": round(latency_pass(cams, n_lat, sb.FLAG_TILE_ORDER), 4),
                                     "orbit_ms": round(latency_pass(orbit, n_lat, sb.FLAG_TILE_ORDER), 4),
                                     "orbit_quarter_degree_ms": round(latency_pass(slow, n_lat, sb.FLAG_TILE_ORDER), 4),
                                     "orbit_quarter_degree_ms_default_order": round(latency_pass(slow, n_lat), 4)}

    wd.phase("warm-up")
    # ---- warm-up, then the timed region ---------------------------------------------------
    for k in range(args.warmup):
        step(k, last=(k == args.warmup - 1))
    drain()
    barrier()
    wd.phase("timed region", quiet=True)              # (no print between the barrier and the clock)
    resent_before = resent                    # (the moving-camera passes above resend tails by design; the timed region should not)
    host_t.update(launch=0.0, gather=0.0, finish=0.0, groups=0)
    del asm_ev[:]
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(k, timed=True, last=(k == args.steps - 1))
    drain()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = counters.cuda() if nccl else counters
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        counters = c.cpu()
    # average duration of the ray-march launch over the timed region (HIP events on the
    # stream each launch went to; with frames in flight the launches overlap each other)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))
    frames_per_launch = float(np.mean([n for _, _, n in ev]))
    # every rank's OWN share time per frame (HIP events around its launches in the timed region: the launches of the groups in flight
    # overlap, so this is a launch's duration, not the rank's throughput) and rank 0's assembly per frame -- one SCALE record shows an
    # imbalance
    per_rank_ms = [round(float(np.sum([a.elapsed_time(b) for a, b, _ in ev])) / max(1, sum(n for _, _, n in ev)), 5)]
    rank0_assemble_ms = (round(float(np.sum([a.elapsed_time(b) for a, b, _ in asm_ev])) / max(1, sum(n for _, _, n in asm_ev)), 5)
                         if (rank == 0 and asm_ev) else None)
    if sharded and world > 1:
        got = [None] * world
        dist.all_gather_object(got, per_rank_ms[0])
        per_rank_ms = got
    # ... and the run's STEADY STATE beside a short run's burst: a short sharded run (tiles.is_burst) deals its bands as that
    # burst and launches in tile order; the same deal rendered for >= 16 fills of the pipeline in the default order is what a long run gets
    steady = None
    if sharded and world > 1 and pt is None and args.orbit == 0:
        n_steady = max(args.steps, 16 * G * nbuf)
        saved_flags = flags
        if plan_ordered and not args.tile_order:
            flags &= ~sb.FLAG_TILE_ORDER
        try:
            steady = {"ms_per_step": round(throughput_pass(cams, n_steady), 5), "steps": n_steady,
                      "shares_in_tile_order": bool(flags & sb.FLAG_TILE_ORDER),
                      "is": "the timed run's deal and buffers over >= 16 fills of the pipeline, in the launch order a long run uses "
                            "(tiles.group_plan): the steady-state time per frame beside a short run's burst figure"}
        finally:
            flags = saved_flags

    wd.phase("check (assembled frames against a whole-frame render)")
    check_ok = None
    # (always when several ranks took part: a scaling curve is worth what its frames are -- one whole-frame render and a few compares)
    if (args.check or world > 1) and sharded and rank == 0:
        refs = {}

        def ref_of(k):                     # the whole-frame render of frame k's camera
            c = k % len(cams)
            if c not in refs:
                ref = torch.zeros((H, W) + px_shape, dtype=px_dtype, device="cuda")
                if pt is not None:
                    scene.DrawPathDevice(cams[c], W, H, ref.data_ptr(), pt=pt, flags=flags, stream=main)
                else:
                    scene.DrawDevice(cams[c], W, H, ref.data_ptr(), flags=flags, stream=main)
                torch.cuda.synchronize()
                refs[c] = ref
            return refs[c]
        # every frame of every group buffer that the timed steps filled
        filled = [(g % nbuf, w, g * G + w) for g in range(max(0, (args.steps - 1) // G + 1 - nbuf), (args.steps - 1) // G + 1)
                  for w in range(G) if g * G + w < args.steps]
        check_ok = all(bool(torch.equal(frame[sl][w].view(torch.int32), ref_of(k).view(torch.int32))) for sl, w, k in filled)

    # The ranks part HERE, before rank 0's report: nothing below is a collective, so a long report (the bandwidth kernels, with
    # --configs all the other configurations) cannot run a peer's last barrier into the process group's timeout.
    if sharded:
        wd.phase("last barrier (every rank is done with its collectives)", limit=args.init_timeout + 60.0)
        dist.barrier()
    wd.phase("report (bandwidth, configs, cpu baseline: rank 0 alone, no collective)", quiet=world == 1)
    if rank == 0:
        sec_per_step = elapsed / args.steps
        copy_gbs = measured_hbm_bandwidth(sb, device)   # the box's own figure beside the nameplate
        mode = "spp%d" % args.spp if pt is not None else "display" if args.display else "compact" if compact else "default"
        if args.shadow_queue and not compact and pt is None and not args.one_kernel:
            mode += ":shadow-queue"
        if args.one_kernel and not compact:
            mode = "one-kernel" if mode == "default" else mode + ":one-kernel"     # k_plain / k_path: other kernels, other counters
        if sharded:
            mode += ":sharded"                         # other kernel instances (sparse shares, bands): no PMC pass of its own
        # ... and of the same grid (SDFHIP_TOP_GRID_LEVEL / _SPLIT change it): "grid9" dense, "grid8+blocks" split
        if args.orbit > 0:
            mode += f":orbit{args.orbit}"                # a camera that moves every frame: counters of its own (profiles/*_orbit_pmc.json)
        key_base, gsuf = f"{W}x{H}:{scene_name}:{mode}", grid_suffix(scene, pt)
        mode += gsuf
        pmc = load_pmc(key_base + gsuf) if world == 1 else None
        headline = (world == 1 and not sharded and (W, H) == (1920, 1080) and args.depth == 9 and not args.asdf and pt is None and not compact
                    and not args.display and not args.one_kernel and not args.shadow_queue and not args.tile_order and args.orbit == 0
                    and args.kernel == "auto")
        # the same counters measured NOW, by child processes under rocprofv3 --pmc (bench_live_pmc.py); the committed pass stays beside them
        live_note = None
        live_state = {"broken": None}

        def live(workload_args, fpl=1.0):
            """one workload's passes; once a pass has HUNG (not merely failed) no further child is started in this run"""
            if live_state["broken"]:
                return {"dropped": "not run: " + live_state["broken"]}
            try:
                r = live_pmc(workload_args, kernel_source_hash(), fpl)
            except Exception as e:         # (a counter file of another shape, a full /tmp: the committed figure stands, the line is printed)
                r = {"dropped": f"{type(e).__name__}: {str(e)[:200]}"}
            if "did not end within" in str(r.get("dropped", "")):
                live_state["broken"] = "an earlier pass of this run hung (" + r["dropped"] + ")"
            return r
        if headline and args.live_pmc != "off" and not args.only_timed and not args.lab:
            wd.phase("live PMC passes (this command again, as children under rocprofv3 --pmc)", quiet=True)
            torch.cuda.synchronize()
            pmc, live_note = merge_pmc(live([], frames_per_launch), pmc)
        live_all = live_note is not None and args.live_pmc in ("auto", "all")
        comp = load_compulsory(key_base + gsuf) if world == 1 else None
        # the headline also reports the counters of the same command under --orbit 90 (a new camera every frame), over that pass's time
        orbit_pmc = None
        if world == 1 and not sharded and args.orbit == 0 and "orbit_ms_per_step" in latency:
            orbit_pmc = {"pmc": load_pmc(f"{key_base}:orbit{len(orbit)}{gsuf}"), "ms_per_step": latency["orbit_ms_per_step"], "cameras": len(orbit)}
            if live_all:
                orbit_pmc["pmc"], orbit_note = merge_pmc(live(["--orbit", str(len(orbit))], frames_per_launch), orbit_pmc["pmc"])
                live_note["orbit"] = {k: v for k, v in orbit_note.items() if k != "passes"}
        roof = roofline(sec_per_step, own_bytes_rank, ref_bytes_rank, pmc, copy_gbs, compulsory=comp, latency_ms=latency["ms"], orbit=orbit_pmc)
        # what the fraction divides by, so that it can be recomputed from profiles/: per_frame / time_ms / peak.  kernel_ms is
        # the HIP-event time around one frame's launches on their stream (k_march; overlapping the other frame in
        # flight), and frac_over_kernel_ms the same fraction over that longer time
        roof.update({"time_ms": round(sec_per_step * 1e3, 4), "kernel_ms": round(kernel_ms, 4), "frames_per_launch": frames_per_launch,
                     "kernel_ms_is": "average HIP-event time around one launch on its stream; with several frames in flight the launches overlap, "
                                     "so this is longer than time_ms (the steady-state time per frame, which the fractions divide by)"})
        if live_note is not None:
            roof["live_pmc"] = live_note
            if isinstance(roof.get("traffic_source"), dict):
                roof["traffic_source"]["live"] = live_note.get("used") == "live"
        # ---- continuous operation: the same frames, pipelined the same way, for seconds; clocks sampled ----
        sus_s = sustained_seconds(args.sustained, headline)
        sustained = telemetry = None
        if any(sus_s) and world == 1 and not sharded and pt is None:
            wd.phase("sustained legs (continuous rendering, clock / power / temperature sampled every 100 ms)", quiet=True)
            telemetry = GpuTelemetry(bus_ids[0])
            valu_pf = pmc.get("valu_insts_per_frame") if isinstance(pmc, dict) else None
            sustained = {"is": "the timed region's frames rendered without a stop for >= `seconds` (Program.cs:58-74 never stops): device time "
                               "from HIP events at chunk boundaries on every stream, nothing synchronised inside a leg; telemetry from "
                               + str(telemetry.source or telemetry.error)}

            def leg(camlist, seconds):
                r = sustained_leg(torch, lambda k, si: render(local[si][0], streams[si].cuda_stream, c=camlist[k % len(camlist)]),
                                  streams, seconds, telemetry)
                clk = ((r.get("telemetry") or {}).get("sclk_mhz") or {}).get("mean")
                r["valu_frac_of_spec_2400mhz"] = (round(valu_pf / (r["ms_per_step"] * 1e-3) / 1e9 / VALU_PEAK_SPEC_GINSTR, 4) if valu_pf else None)
                r["valu_frac_at_observed_clock"] = at_observed_clock(valu_pf, r["ms_per_step"], clk)
                r["mray_per_s"] = round(W * H / (r["ms_per_step"] * 1e-3) / 1e6, 1)
                return r
            try:
                if sus_s[0] > 0:
                    sustained["cfg2"] = leg([cam], sus_s[0])
                if sus_s[2] > 0:
                    sustained["cfg2_orbit"] = leg(orbit, sus_s[2])
            except Exception as e:
                sustained["error"] = f"{type(e).__name__}: {e}"
            flat = sustained.get("cfg2")
            if flat:                               # flat scalars in `roofline`: the driver's record keeps that object's scalars
                tl = flat.get("telemetry") or {}
                roof.update({"sustained_ms_per_step": flat["ms_per_step"], "sustained_seconds": flat["seconds"], "sustained_frames": flat["frames"],
                             "sustained_ms_per_step_first_20": flat["ms_per_step_first_20"],
                             "sustained_ms_per_step_last_1000": flat["ms_per_step_last_1000"],
                             "sustained_sclk_mhz_min": (tl.get("sclk_mhz") or {}).get("min"),
                             "sustained_sclk_mhz_mean": (tl.get("sclk_mhz") or {}).get("mean"),
                             "sustained_power_w_mean": (tl.get("power_w") or {}).get("mean"),
                             "sustained_temp_c_max": (tl.get("temp_c") or {}).get("max"),
                             "sustained_valu_frac_of_spec": flat["valu_frac_of_spec_2400mhz"],
                             "sustained_valu_frac_at_observed_clock": flat["valu_frac_at_observed_clock"]})
        out = {
            "metric": "Mray/s (primary rays; frame W*H / time per frame)",
            "value": round(W * H * max(1, args.spp) / sec_per_step / 1e6, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(sec_per_step * 1e3, 4),
            "latency_ms": latency["ms"],
            # the same frame as a viewer would ask for it (SDFHIP_FLAG_TILE_ORDER: tiles in the order of the last frame's cost), at rest
            # and with the camera moving one degree per frame; null where the flag does not apply (sharded, compact, one-kernel runs)
            "latency_ms_tile_order": latency["tile_order"]["ms"] if "tile_order" in latency else None,
            "latency_ms_tile_order_moving_camera": latency["tile_order"]["orbit_ms"] if "tile_order" in latency else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"{W}x{H} path trace, {args.spp} spp, 3 diffuse bounces, seed 0x5DFB0C5, " if pt is not None else
                             f"{W}x{H} primary-ray sphere trace + shadow march, ") + f"{scene_name} "
                            f"(N={od.Length} nodes, {od.nbytes / 1e6:.1f} MB), camera (0.5,0.5,-0.35) yaw 0.35 pitch -0.2" +
                            (f", moving 1 degree per frame round the scene ({len(cams)} cameras)" if args.orbit > 0 else ""),
                "kernel": ("path/" if pt is not None else "") +
                          ("stack" if (kernel_used & 0xF) == sb.KERNEL_STACK else "generic") + ("+compact" if compact else "") +
                          ("" if pt is not None else
                           ", k_march (primary march, shading; the shadow rays of waves that hold fewer than 32 queued by ballot + prefix) -> k_shadow"
                           if (compact and (kernel_used & 0xF) == sb.KERNEL_STACK) else ", k_compact: persistent waves, lane refill" if compact else
                           ", one kernel" if args.one_kernel else
                           ", k_march (primary march, shading) -> queue (wavefront ballot + prefix compaction of every shadow ray) -> k_shadow"
                           if args.shadow_queue else
                           ", k_march: primary march, shading and shadow march as three wave-converged loops of one kernel"),
                "top_grid": {"level": scene.top_grid_level, "bytes": scene.top_grid_bytes},
                # BASELINE cfg-3 says "wavefront ray compaction on": which form this line is (three are built, all measured slower than none)
                "compaction": ("the shadow rays of waves holding fewer than 32 compacted by ballot/prefix into a queue, marched 64 to a wave by k_shadow "
                               "(SDFHIP_FLAG_COMPACT)" if (kernel_used & 0xF) == sb.KERNEL_STACK else
                               "persistent waves, ballot/prefix lane refill (k_compact: SDFHIP_FLAG_COMPACT on a tree without a full-depth grid)") if compact else
                              "every shadow ray compacted by ballot/prefix into a queue, marched by k_shadow (laboratory)" if args.shadow_queue else "off",
                "parallelism": "1 GPU" if not sharded else
                               f"{world} GPU(s), {args.band_rows}-row bands " +
                               ("round-robin" if not layout.weighted else
                                f"dealt by measured cost, rank 0 charged for the assembly (rank 0: {len(layout.bands_of(0))} of {layout.n_bands} bands)"
                                if layout.dealt_by_cost else
                                f"dealt by weight (rank 0: {layout.rank0_weight:.3f} of a peer's share)") +
                               f" + gather to rank 0 ({args.backend})",
                "ranks_seen": dist.get_world_size() if sharded else 1,
                "pci_bus_ids": bus_ids, "distinct_gpus": len(set(bus_ids)),
                "transport": (("RCCL through torch.distributed (backend nccl)" if nccl else "gloo through host buffers (a rehearsal, not a performance mode)")
                              if sharded else None),
                "frames_in_flight": nbuf * G,
                # the hardware queues the runtime deals this process's streams onto: read when the runtime starts, so what the
                # ENVIRONMENT held then (a PMC pass must be collected with the same value: scripts/profile.sh exports it)
                "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "frames_per_gather": G if sharded else None,
                "frames_rendered_in_the_timed_region": rendered,
                "gather_pixel_bytes": (round(s2["prefix"] / (rows_local * W * G), 3) if sparse2 else px_bytes) if sharded else None,
                "gather_format": ("sparse shares written by the march kernel" if sparse2 else "frame pixels") if sharded else None,
                "shares_in_tile_order": bool(flags & sb.FLAG_TILE_ORDER) if sharded else None,
                # rank 0's host time per gather group in the timed region, by part (the sparse-share pipeline): the launch call, issuing the
                # gather (+ a peer's counter copy), and finish() -- the wait for the slot's previous group, its expansion call, the counters
                # what a short run gets that a long one does not, from ONE predicate (tiles.is_burst): tile order on the batched launches,
                # the band deal timed as that burst
                "plan": ({"frames_per_gather": G, "groups_in_flight": nbuf, "burst": bool(plan_ordered),
                          "band_deal_timed_as": (None if deal is None else f"{args.steps}-step burst" if plan_ordered else "steady state"),
                          "shares_in_tile_order": bool(flags & sb.FLAG_TILE_ORDER)} if sharded else None),
                "per_rank_ms": per_rank_ms if sharded else None,
                "per_rank_ms_is": ("each rank's own launches in the timed region, HIP-event time per frame (groups in flight overlap: a launch's "
                                   "duration, not a throughput)") if sharded else None,
                "rank0_assemble_ms": rank0_assemble_ms,
                "links": links,
                "host_us_per_group": ({k: round(host_t[k] / max(1, host_t["groups"]) * 1e6, 1) for k in ("launch", "gather", "finish")}
                                      if sparse2 and host_t["groups"] else None),
                "float_tails_sent_again": resent if sparse2 else None,
                "float_tails_sent_again_in_the_timed_region": (resent - resent_before) if sparse2 else None,
                "output": "RGBA8, display pass fused (DisplayFrag.hlsl)" if args.display else "RGBA32F, alpha = step count",
                "gstep_per_s": round(float(counters[2]) / sec_per_step / 1e9, 3),
                "shadow_rays_per_frame": int(counters[3]),
                "scene_build_s": round(t_gen, 2),
            },
            # latency: host clock around launch + completion of ONE frame (sharded: one gather group of `frames`
            # frames) with nothing else in flight -- what an interactive viewer waits for; orbit_*: the same, and the
            # pipelined time per frame, with the camera moving every frame (caches see a new access pattern each time)
            "latency": latency,
            "roofline": roof,
        }
        if steady is not None:
            steady["value"] = round(W * H * max(1, args.spp) / (steady["ms_per_step"] * 1e-3) / 1e6, 2)
            out["steady_state"] = steady
        if check_ok is not None:
            out["config"]["assembled_frame_equals_whole_frame_render"] = check_ok
        if sustained is not None:
            out["sustained"] = sustained
        if args.configs == "all" or (args.configs == "auto" and headline):
            # (the headline's buffers are not needed any more: the path-traced configuration wants 30 GB of queues)
            local = send = frame = gathered = None
            torch.cuda.empty_cache()
            # (a configuration that fails says so in its own entry: the headline above has been measured and is printed regardless)
            try:
                out["configs"] = run_configs(sb, torch, scene, scene_name, copy_gbs, args.configs_scale, args.depth, streams,
                                             sustained_4k_seconds=sus_s[1] if sustained is not None else 0.0, telemetry=telemetry,
                                             live=live if live_all else None)
            except Exception as e:
                out["configs"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(od, cam, W, H, args.cpu_seconds)
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        # LAST key (the driver's record keeps the line's last 2 000 characters): every configuration in <= 600 characters
        out["configs_summary"] = configs_summary(out, out.get("configs"))
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    scene.close()
    if sharded:
        dist.destroy_process_group()


# ---- the CPU baseline: the ONE leg of the bench that may touch oracle/ (it is the thing timed here, never the thing shipped) ----
def cpu_limits():
    """What this process may use of the host: the CPUs of its affinity mask and the cgroup's CPU quota (cpu.max of cgroup v2, or
    cfs_quota_us / cfs_period_us of v1) in CPUs -- None when there is no quota."""
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    return affinity, quota


def cpu_baseline(od, cam, W, H, target_seconds):
    """The CPU oracle (the build's C restatement of Compute.hlsl: the reference has no CPU path) timed on this
    host over bounded samples of the same frame (every `step`-th row), built here with -O3 -march=native -- same
    source, same -ffp-contract=off, its rows compared bit for bit with the portable build the parity tests use.  Timed by
    oracle_bench_rows: a pool of threads pinned to the CPUs of this process's affinity mask, dealt over the NUMA nodes, each
    node reading its own copy of the scene, all waiting at a barrier before the clock (read inside the C function) starts;
    pixels dealt in chunks of 64 from one counter.  The thread count is SWEPT (1, 16, 64, 128, 256, the affinity mask's size and
    the cgroup quota, whichever the mask allows) and the best is the reported value; `limits` says what the host let this
    process use -- a quota of 16 CPUs caps every thread count at 16 CPUs' worth of cycles."""
    import oracle
    oracle.build()
    native = oracle.build_native()                     # None when the host has no compiler: then only the portable build is timed
    affinity, quota = cpu_limits()
    usable = min(affinity, int(quota + 0.999)) if quota else affinity
    cands = sorted({n for n in (1, 16, 64, 128, 256, affinity, usable) if 1 <= n <= affinity})
    # calibrate one thread on a sparse sample of the frame (rows spread top to bottom), on both builds: the faster one is swept
    # (-O3 -march=native is not always it; the other's one-thread figure is printed beside it)
    step0 = max(1, H // 8)
    builds = {"gcc -O2 -march=x86-64-v2 (+ an fma clone) -ffp-contract=off": False}
    if native is not None:
        builds["gcc -O3 -march=native -ffp-contract=off, built on this host"] = True
    calib = {}
    for name, nat in builds.items():
        _, _, sec0, _ = oracle.bench_rows(od.Structs, od.Values, cam.State, W, H, row_step=step0, nthreads=1, store=False, native=nat)
        calib[name] = sec0 / ((H + step0 - 1) // step0)
    build_used = min(calib, key=calib.get)
    kw = {"native": builds[build_used]}
    per_row1 = calib[build_used]
    budget = target_seconds / len(cands)
    sweep, best, best_img, best_step = [], None, None, 1
    for nt in cands:
        # rows for ~budget seconds if the threads scaled perfectly up to what the host lets this process use; when the whole frame
        # is not enough (a cgroup grants CPU time in 100 ms periods: a quota shows in a timed region of seconds, not in a burst of
        # 0.2 s) the frame is rendered several times over
        want = budget * min(nt, usable) / max(per_row1, 1e-9)
        rows = int(min(H, max(8, want)))
        step = max(1, H // rows)
        nrows = (H + step - 1) // step
        repeat = max(1, int(round(want / nrows)))
        img, _, sec, topo = oracle.bench_rows(od.Structs, od.Values, cam.State, W, H, row_step=step, nthreads=nt, repeat=repeat, **kw)
        e = {"threads": topo["threads"], "value": round(repeat * nrows * W / sec / 1e6, 3), "numa_nodes": topo["numa_nodes"],
             "scene_copies": topo["scene_copies"],
             "sample": f"every {step}th row = {nrows * W} pixels" + (f", {repeat} times over," if repeat > 1 else "") + f" in {sec:.2f} s"}
        sweep.append(e)
        if best is None or e["value"] > best["value"]:
            best, best_img, best_step = e, img, step
    other = {name: {"one_thread_ms_per_row": round(v * 1e3, 3)} for name, v in calib.items()}
    same = None
    if kw["native"]:                                   # a few of the best run's rows through the portable build: identical bits
        nrows = best_img.shape[0]
        k = max(1, nrows // 8)
        ref, _ = oracle.render(od.Structs, od.Values, cam.State, W, H, row0=0, nrows=(nrows + k - 1) // k, row_step=best_step * k,
                               nthreads=min(usable, 32))
        a, b = best_img[::k].view(np.uint32), ref.view(np.uint32)
        same = bool(((a == b) | (np.isnan(best_img[::k]) & np.isnan(ref))).all())
    one = next(e for e in sweep if e["threads"] == 1)
    model, physical = host_cpu()
    ratio = best["value"] / one["value"] if one["value"] else None
    return {
        "value": best["value"],
        "unit": "Mray/s",
        "cores": best["threads"],
        "kind": "port",
        "cpu_model": model,
        "physical_cores": physical,
        "limits": {"affinity_cpus": affinity, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
                   "note": ("the cgroup lets this process use %.1f CPUs' worth of cycles: more threads than that share them" % quota)
                           if quota and quota < affinity else None},
        "build": build_used,
        "builds_tried_one_thread": other,
        "native_build_equals_portable_build": same,
        "sample": best["sample"] + f" of the same {W}x{H} frame; oracle/sdf_oracle.c::oracle_bench_rows, {best['threads']} pinned pthreads over "
                  f"{best['numa_nodes']} NUMA node(s) ({best['scene_copies']} node-local scene copies), started before the clock, 64-pixel chunks from one counter",
        "sweep": sweep,
        "speedup_over_one_thread": round(ratio, 1) if ratio else None,
        "one_thread": {"value": one["value"], "unit": "Mray/s", "sample": one["sample"]},
    }


def host_cpu():
    """(model name, physical cores) from /proc/cpuinfo; (None, None) when it cannot be read."""
    try:
        model, cores = None, set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model is None:
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None and core is not None:
                    cores.add((phys, core)); phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
        return model, (len(cores) or None)
    except OSError:
        return None, None


if __name__ == "__main__":
    main()
