"""bench_sharded.py -- the parts of bench.py that exist for N > 1: the watchdog of a first contact, the searches that deal the frame's
bands before anything is timed (measure_rank0_weight, measure_band_deal), the launcher of `python bench.py --gpus N` without
torch.distributed.run (spawn_ranks) and the single-process A/B through the library's own multi-device entry points
(main_single_process).  Split out of bench.py in round 6 without a change of behaviour."""
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))

from bench_configs import load_package, orbit_cameras
from bench_report import measured_hbm_bandwidth, roofline


class Watchdog:
    """First contact with N GPUs must end in bounded time and say where it stopped: a daemon thread that
    writes `[bench rank r] phase ..., s in it` to stderr every `every` seconds while the run is in a phase that can block on
    another rank, and ends THIS process with os._exit(3) -- a fresh exit, no re-exec, no clean-up that could block in turn --
    once the run has taken `budget` seconds (or a phase its own limit).  The launcher then ends the other ranks."""

    def __init__(self, rank, budget, every=5.0):
        import threading
        self.rank, self.budget, self.every = rank, budget, every
        self.t0 = self.t_phase = time.monotonic()
        self.name, self.limit, self.quiet = "start", None, True
        self.lock = threading.Lock()
        self.thread = None
        if budget > 0:
            self.thread = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
            self.thread.start()

    def phase(self, name, limit=None, quiet=False):
        """enter a phase; `limit` = seconds this phase alone may take; quiet phases are not logged while they run"""
        with self.lock:
            self.name, self.limit, self.quiet, self.t_phase = name, limit, quiet, time.monotonic()
        if self.budget > 0 and not quiet:
            print(f"[bench rank {self.rank}] phase: {name}", file=sys.stderr, flush=True)

    def _run(self):
        while True:
            time.sleep(min(self.every, 1.0))
            now = time.monotonic()
            with self.lock:
                name, limit, quiet, in_phase = self.name, self.limit, self.quiet, now - self.t_phase
            total = now - self.t0
            over = total > self.budget or (limit is not None and in_phase > limit)
            if over:
                why = (f"phase '{name}' has taken {in_phase:.0f} s (limit {limit:.0f})" if (limit is not None and in_phase > limit)
                       else f"the run has taken {total:.0f} s (budget {self.budget:.0f}), in phase '{name}' for {in_phase:.0f} s")
                print(f"[bench rank {self.rank}] WATCHDOG: {why}: giving up with exit code 3", file=sys.stderr, flush=True)
                os._exit(3)
            if not quiet and in_phase >= self.every and int(in_phase / self.every) != int((in_phase - min(self.every, 1.0)) / self.every):
                print(f"[bench rank {self.rank}] still in phase '{name}' after {in_phase:.0f} s ({total:.0f} s of {self.budget:.0f})",
                      file=sys.stderr, flush=True)


def main_single_process(args, json_fd):
    """`--single-process`: the N-device frame behind the library's one call (sdfhip_multi_submit / _wait): one process, one
    host thread and one stream per device inside libsdfhip.so, sparse shares written by the march kernel, pushed into
    device 0 over the peers' own links, assembled there.  Same workload, same JSON line; `config.parallelism` says which
    of the two things it measures."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")           # (see main())
    import torch

    sb = load_package(args)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    ndev = torch.cuda.device_count()
    if max(devices) >= ndev:
        raise SystemExit(f"--devices {devices}: this box has {ndev} GPU(s)")
    W, H = (int(v) for v in args.size.lower().split("x"))
    t0 = time.time()
    if args.asdf:
        od = sb.OctData.LoadAsdf(args.asdf)
        scene_name = os.path.basename(args.asdf)
    else:
        od = sb.dragon_standin(args.depth, nthreads=max(1, min(32, os.cpu_count() or 1)))
        scene_name = f"dragon_standin_d{args.depth}"
    t_gen = time.time() - t0
    torch.cuda.set_device(devices[0])
    ms = sb.MultiScene(od, devices)                  # (create runs sdfhip_multi_selftest: a link that does not deliver fails here)
    links = ms.selftest()
    if args.band_rows != 16 or args.rank0_weight > 0:
        ms.configure(band_rows=args.band_rows, rank0_weight=args.rank0_weight if args.rank0_weight > 0 else 1.0)
    cam = sb.Logic(W, H)
    cam.Position = (0.5, 0.5, -0.35)
    cam.Heading = (-0.2, 0.35)
    cams = orbit_cameras(sb, W, H, args.orbit) if args.orbit > 0 else [cam]
    pt = sb.PathTrace(spp=args.spp) if args.spp > 0 else None
    flags = (sb.FLAG_DISPLAY if args.display else 0) | (sb.FLAG_TILE_ORDER if args.tile_order else 0)
    frame_mode = args.multi_mode == "frame" or pt is not None
    G = 1 if frame_mode else (args.gather_every if args.gather_every > 0 else (8 if len(devices) >= 8 else 4))
    nslots = 1 if frame_mode else min(4, args.frames_in_flight if args.frames_in_flight > 0 else 4)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    inflight = [None] * nslots              # the step index of the group a slot holds
    stats_seen = []

    def run(n, collect=False):
        """n steps (frames) through the slots; returns the host time"""
        for k in range(nslots):
            if inflight[k] is not None:
                ms.Wait(k); inflight[k] = None
        sync_all()
        t = time.perf_counter()
        k = 0
        while k < n:
            g = min(G, n - k)
            slot = (k // G) % nslots
            if inflight[slot] is not None:
                _, st = ms.Wait(slot, want_stats=True)
                if collect:
                    stats_seen.append(st)
            ms.Submit(slot, [cams[(k + i) % len(cams)] for i in range(g)], W, H, flags=flags, pt=pt)
            inflight[slot] = k
            if frame_mode:
                _, st = ms.Wait(slot, want_stats=True)
                inflight[slot] = None
                if collect:
                    stats_seen.append(st)
            k += g
        for q in range(nslots):
            if inflight[q] is not None:
                _, st = ms.Wait(q, want_stats=True)
                inflight[q] = None
                if collect:
                    stats_seen.append(st)
        sync_all()
        return time.perf_counter() - t

    run(args.warmup if args.warmup > 0 else 1)
    elapsed = run(args.steps, collect=True)
    # one frame alone across the devices: submit + wait, nothing else in flight (median)
    lat = []
    for j in range(20 if pt is None else 3):
        sync_all()
        t = time.perf_counter()
        ms.Submit(0, cams[j % len(cams)], W, H, flags=flags, pt=pt)
        ms.Wait(0)
        lat.append(time.perf_counter() - t)
    latency_ms = float(np.median(lat)) * 1e3
    # the frame to a HOST array, one call (what the C# host makes): sdfhip_multi_render
    host = np.empty((H, W, 4), dtype=np.uint8 if args.display else np.float32)
    hl = []
    for j in range(8 if pt is None else 2):
        t = time.perf_counter()
        ms.Draw(cams[j % len(cams)], W, H, flags=flags, pt=pt, out=host)
        hl.append(time.perf_counter() - t)
    host_ms = float(np.median(hl[1:])) * 1e3

    check_ok = None
    one = sb.Scene(od, device=devices[0])
    st = sb.Stats()
    ref = torch.zeros((H, W) if args.display else (H, W, 4), dtype=torch.int32 if args.display else torch.float32, device=f"cuda:{devices[0]}")
    main_stream = torch.cuda.current_stream().cuda_stream
    if pt is not None:
        one.DrawPathDevice(cam, W, H, ref.data_ptr(), pt=pt, flags=sb.FLAG_COUNT, stream=main_stream, stats=st)
    else:
        one.DrawDevice(cam, W, H, ref.data_ptr(), flags=sb.FLAG_COUNT, stream=main_stream, stats=st)
    torch.cuda.synchronize()
    if args.check:
        check_ok = True
        for j in range(min(len(cams), 6)):
            if pt is not None:
                one.DrawPathDevice(cams[j], W, H, ref.data_ptr(), pt=pt, stream=main_stream)
            else:
                one.DrawDevice(cams[j], W, H, ref.data_ptr(), flags=flags & ~sb.FLAG_TILE_ORDER, stream=main_stream)
            torch.cuda.synchronize()
            ms.Draw(cams[j], W, H, flags=flags, pt=pt, out=host)
            check_ok = check_ok and bool(np.array_equal(host.view(np.uint32 if not args.display else np.uint8),
                                                        ref.cpu().numpy().view(np.uint32 if not args.display else np.uint8).reshape(host.shape)))
    sec_per_step = elapsed / args.steps
    px_bytes = 4 if args.display else 16
    ref_bytes = 8 * st.n_nodes + 8 * st.n_samples + px_bytes * W * H
    own_bytes = 16 * st.n_loads + px_bytes * W * H
    roof = roofline(sec_per_step, own_bytes, ref_bytes, None, measured_hbm_bandwidth(sb, devices[0]))
    roof.update({"time_ms": round(sec_per_step * 1e3, 4), "note": "no PMC pass exists for a multi-device run: no fraction, only the demand figures"})
    n_st = max(1, len(stats_seen))
    out = {
        "metric": "Mray/s (primary rays; frame W*H / time per frame)",
        "value": round(W * H * max(1, args.spp) / sec_per_step / 1e6, 2),
        "unit": "Mray/s",
        "n_gpus": len(devices),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(sec_per_step * 1e3, 4),
        "latency_ms": round(latency_ms, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"{W}x{H} path trace, {args.spp} spp, 3 diffuse bounces, seed 0x5DFB0C5, " if pt is not None else
                         f"{W}x{H} primary-ray sphere trace + shadow march, ") + f"{scene_name} (N={od.Length} nodes, {od.nbytes / 1e6:.1f} MB), "
                        "camera (0.5,0.5,-0.35) yaw 0.35 pitch -0.2" + (f", moving 1 degree per frame ({len(cams)} cameras)" if args.orbit > 0 else ""),
            "parallelism": f"single process, devices {devices} through sdfhip_multi_submit/_wait (one host thread + stream per device, "
                           f"{args.band_rows}-row bands, sparse shares written by the march kernel, gather by {ms.transport} into device {devices[0]})",
            "ranks_seen": len(links), "pci_bus_ids": [l["pci_bus_id"] for l in links], "distinct_gpus": len({l["pci_bus_id"] for l in links}),
            "transport": "RCCL ncclSend / ncclRecv inside the library" if ms.transport == "rccl" else "hipMemcpyPeerAsync on the senders' streams",
            "links": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in l.items()} for l in links],
            "measures": ("one frame at a time across all devices, submit to completion: the viewer's latency (strong scaling of ONE frame)"
                         if frame_mode else
                         f"throughput of groups: {G} frames per launch and gather, {nslots} groups in flight ({G * nslots} frames in flight)"),
            "frames_per_gather": G, "groups_in_flight": nslots,
            "gathered_bytes_per_frame": round(sum(s.gathered_bytes for s in stats_seen) / n_st / G, 1),
            "float_tails_sent_again": int(sum(s.resends for s in stats_seen)),
            "rank_ms_per_group": [round(sum(s.rank_ms[r] for s in stats_seen) / n_st, 4) for r in range(len(devices))],
            "host_frame_ms": round(host_ms, 4),
            "output": "RGBA8, display pass at assembly" if args.display else "RGBA32F, alpha = step count",
            "scene_build_s": round(t_gen, 2),
        },
        "latency": {"frames": 1, "ms": round(latency_ms, 4), "to_host_array_ms": round(host_ms, 4)},
        "roofline": roof,
    }
    if check_ok is not None:
        out["config"]["assembled_frame_equals_whole_frame_render"] = check_ok
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    one.close()
    ms.close()


LINK_CHECK_EXIT = 5


def link_check(torch, dist, rank, world, nccl, wd, names, limit=10.0, nbytes=1 << 20, peer_access=None):
    """First contact with the LINKS, before anything depends on them (the library's own design has the same step
    at create: sdfhip_multi_selftest).  Every peer sends rank 0 one `nbytes` message of a pattern that only it makes (the
    direction every gather and every resent tail takes: over xGMI each peer has its own link into rank 0); rank 0 waits at most
    `limit` seconds per link, compares every byte, prints a per-link table on stderr and tells everyone the verdict in a broadcast.
    A link that is silent or delivers other bytes ends EVERY rank non-zero (exit code 5) with the link named; a rank that waits
    for a verdict which never comes is ended by the watchdog (exit code 3) with this phase in its message.
    names[r]: how rank r is called in the table (its PCI bus id, or 'host buffers' in the gloo rehearsal).  peer_access[r]: what
    hipDeviceCanAccessPeer said about r's device and rank 0's (None: not asked).  -> the table's rows (rank 0) or None."""
    import datetime
    dev = "cuda" if nccl else "cpu"
    # test hook (tests/test_first_contact.py): 'corrupt:R' -- rank R's payload is damaged on the way; 'silent:R' -- rank R never sends
    fault = os.environ.get("SDFHIP_BENCH_LINK_FAULT", "")
    fkind, _, frank = fault.partition(":")
    frank = int(frank) if frank.strip().isdigit() else -1

    def pattern(r):
        i = torch.arange(nbytes, dtype=torch.int64, device=dev)
        return ((i * 131 + r * 17 + (i >> 9)) % 251).to(torch.uint8)

    def name(r):
        return f"rank {r} ({names[r]})"
    verdict = torch.zeros(world, dtype=torch.int32, device=dev)          # 0 ok, 1 other bytes, 2 nothing within the limit
    rows = None
    if rank == 0:
        rows = []
        for r in range(1, world):
            wd.phase(f"link check: {name(r)} -> {name(0)}, {nbytes >> 10} KB, at most {limit:.0f} s", limit=limit + 20.0)
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            t0 = time.perf_counter()
            work = dist.irecv(buf, src=r)
            done = False
            if nccl:
                # A wait with a timeout does not bound a receive that runs on the GPU.  work.wait() only puts the receive in front of
                # this thread's stream; an EVENT recorded behind it completes when the bytes are there, and querying an event never
                # blocks -- so the limit holds whatever the state of the link (a peer that never even joins the transfer leaves this
                # rank inside irecv: the watchdog's limit on this phase ends that case, naming the link).
                work.wait()
                arrived = torch.cuda.Event()
                arrived.record()
                while time.perf_counter() - t0 < limit:
                    if arrived.query():
                        done = True
                        break
                    time.sleep(0.0005)
            else:                                                          # (gloo's receive completes inside wait(): the timeout goes there)
                try:
                    work.wait(datetime.timedelta(seconds=limit))
                    done = True
                except Exception:
                    done = False
            ms = (time.perf_counter() - t0) * 1e3
            row = {"link": f"{name(r)} -> {name(0)}", "peer_access": None if peer_access is None else peer_access[r], "ms": round(ms, 3),
                   "gb_per_s": round(nbytes / (ms * 1e-3) / 1e9, 3) if done else None, "verdict": "ok"}
            if not done:
                row["verdict"] = f"NOTHING ARRIVED within {limit:.0f} s"
                rows.append(row)
                _print_links(rows)
                print(f"[bench rank 0] link check FAILED: {row['link']}: {row['verdict']} (exit code {LINK_CHECK_EXIT}; the other ranks are "
                      "ended by the launcher or by their own watchdog)", file=sys.stderr, flush=True)
                os._exit(LINK_CHECK_EXIT)                                   # (the process group is wedged: no collective can tell the others)
            if not bool(torch.equal(buf, pattern(r))):
                bad = int((buf != pattern(r)).sum().item())
                row["verdict"] = f"{bad} of {nbytes} bytes differ from what rank {r} sent"
                verdict[r] = 1
            rows.append(row)
        _print_links(rows)
    else:
        wd.phase(f"link check: {name(rank)} -> {name(0)} (send, then rank 0's verdict)", limit=limit * world + 30.0)
        if not (fkind == "silent" and frank == rank):
            payload = pattern(rank)
            if fkind == "corrupt" and frank == rank:
                payload[nbytes // 3] ^= 0x40
            dist.send(payload, dst=0)
            if nccl:
                torch.cuda.synchronize()
    dist.broadcast(verdict, src=0)
    bad = [r for r, v in enumerate(verdict.tolist()) if v]
    if bad:
        for r in bad:
            print(f"[bench rank {rank}] link check FAILED: {name(r)} -> {name(0)} delivered other bytes than were sent: every rank ends "
                  f"with exit code {LINK_CHECK_EXIT}", file=sys.stderr, flush=True)
        os._exit(LINK_CHECK_EXIT)
    return rows


def _print_links(rows):
    print("[bench rank 0] link check (one message from every peer into rank 0, every byte compared):", file=sys.stderr)
    for r in rows:
        pa = "" if r["peer_access"] is None else f", peer access {'yes' if r['peer_access'] else 'NO (copies stage through the host)'}"
        rate = "" if r["gb_per_s"] is None else f" = {r['gb_per_s']} GB/s incl. the channel's set-up"
        print(f"[bench rank 0]   {r['link']}{pa}: {r['ms']} ms{rate}: {r['verdict']}", file=sys.stderr)
    sys.stderr.flush()


def measure_rank0_weight(sb, scene, cam, W, H, world, band_rows, flags, share_shape, px_dtype, px_bytes, rank, nccl, pt, compact,
                         G, nbuf, sparse2=False):
    """Rank 0 also assembles the frame (de-interleave, or the expansion of all ranks' sparse shares), so an
    even deal makes it the slowest rank.  Before anything is timed, rank 0 tries layouts that give it
    0.3 .. 1.0 of a peer's share: for each it times its own work (render + assembly) and the largest
    peer share (rank 1's, which it can render itself: the scene is replicated), and every rank then
    receives the weight with the smallest max of the two."""
    import torch
    import torch.distributed as dist
    T = sb.tiles
    w = torch.ones(1, dtype=torch.float64)
    if rank == 0:
        streams = [torch.cuda.Stream() for _ in range(nbuf)]
        full_shape, full_dtype = ((), torch.int32) if px_bytes == 4 else ((4,), torch.float32)
        frame = torch.zeros((G, H, W) + full_shape, dtype=full_dtype, device="cuda")
        n = 3 if pt is not None else 16

        def work(lay, r, local, gathered):
            def one(k):
                s = streams[k % nbuf].cuda_stream
                if sparse2:          # the march kernel writes the share; rank 0 expands `world` of them (its own stands in for the peers')
                    share = local[k % nbuf]
                    with torch.cuda.stream(streams[k % nbuf]):
                        share[:4].zero_()
                    T.render_sparse2(scene, [cam] * G, W, lay, r, share.data_ptr(), lay.rows_per_rank * W * G, 0, flags=flags, stream=s)
                    if r == 0:
                        T.deinterleave_sparse2(torch.cuda.current_device(), [share.data_ptr()] * world, frame.data_ptr(), W, lay,
                                               lay.rows_per_rank * W * G, frames=G, stream=s)
                    return
                if pt is None and not compact:
                    T.render_bands_batch(scene, [cam] * G, W, lay, r, local[k % nbuf].data_ptr(), flags=flags, stream=s)
                else:
                    for f in range(G):
                        T.render_bands(scene, cam, W, lay, r, local[k % nbuf][f].data_ptr(), flags=flags, stream=s, pt=pt)
                if r == 0:
                    T.deinterleave(torch.cuda.current_device(), gathered.data_ptr(), frame.data_ptr(), W, lay,
                                   stream=s, pixel_bytes=px_bytes, frames=G)
            best = 1e9
            for _ in range(2):
                one(0); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(n):
                    one(k)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (n * G))
            return best

        tried = []
        for cand in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5, 0.4, 0.3):
            lay = T.BandLayout(H, world, band_rows, cand)
            if sparse2:
                local = [torch.zeros(T.sparse2_bytes(lay.rows_per_rank, W, G, lay.rows_per_rank * W * G), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
                gathered = torch.zeros(1, dtype=torch.uint8, device="cuda")
            else:
                local = [torch.zeros((G,) + share_shape(lay.rows_per_rank), dtype=px_dtype, device="cuda") for _ in range(nbuf)]
                gathered = torch.zeros((world, G) + share_shape(lay.rows_per_rank), dtype=px_dtype, device="cuda")
            t0, t1 = work(lay, 0, local, gathered), work(lay, 1, local, gathered)
            tried.append((max(t0, t1), cand, t0, t1))
            del local, gathered
            if t0 <= t1:                               # rank 0 is no longer the slowest: a smaller share only loads the peers
                break
        _, best, t0, t1 = min(tried)
        w[0] = best
        print("[bench] rank-0 share search: " + ", ".join(f"w={c:.2f}: rank0 {a * 1e3:.4f} / peer {b * 1e3:.4f} ms" for _, c, a, b in tried)
              + f" -> {best:.2f}", file=sys.stderr)
        torch.cuda.empty_cache()
    if nccl:
        w = w.cuda()
    dist.broadcast(w, src=0)
    return float(w.item())


def measure_band_deal(sb, scene, cam, W, H, world, band_rows, flags, rank, nccl, G, nbuf, steps, burst=None):
    """The frame's bands dealt by their measured COST (tiles.balanced_owner), with rank 0 -- which also expands all shares into
    the frame -- charged for that work.  Before anything is timed, rank 0 renders the frame once, prices every band from the
    step counts (tiles.band_costs), and tries deals that charge it 0 .. 20 % of the frame's cost for the assembly: for each it
    times its own job (its share + the expansion of `world` shares, its own standing in for the peers') and two peers' shares
    (the scene is replicated: it can render them itself), in the shape the run will have (a short run -- tiles.is_burst: the driver's
    scaling run times 20 steps -- is timed as that burst; the same predicate puts its launches in tile order).  Every rank then receives the deal with the smallest maximum.  -> owner[band]."""
    import torch
    import torch.distributed as dist
    T = sb.tiles
    n_bands = (H + band_rows - 1) // band_rows
    owner = torch.zeros(n_bands, dtype=torch.uint8)
    if rank == 0:
        streams = [torch.cuda.Stream() for _ in range(nbuf)]
        whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        scene.DrawDevice(cam, W, H, whole.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        costs = T.band_costs(whole[..., 3], band_rows)
        del whole
        total = sum(costs)
        frames = torch.zeros((G, H, W, 4), dtype=torch.float32, device="cuda")
        if burst is None:                                     # (one predicate for "a short run": tiles.is_burst, also behind the tile order)
            burst = T.is_burst(world, steps)
        n_frames = steps if burst else 16 * G

        def job(lay, r, shares, expand):
            full = lay.rows_per_rank * W * G

            def run():
                for sh in shares:
                    sh[:4].zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k = 0
                while k < n_frames:
                    g = min(G, n_frames - k)
                    slot = (k // G) % nbuf
                    if k >= G * nbuf:                         # (the share's counter starts at 0 again: its floats are stored as in the run)
                        with torch.cuda.stream(streams[slot]):
                            shares[slot][:4].zero_()
                    T.render_sparse2(scene, [cam] * g, W, lay, r, shares[slot].data_ptr(), full, 0, flags=flags, stream=streams[slot].cuda_stream)
                    if expand:
                        T.deinterleave_sparse2(torch.cuda.current_device(), [shares[slot].data_ptr()] * world, frames.data_ptr(), W, lay, full,
                                               frames=g, stream=streams[slot].cuda_stream)
                    k += g
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n_frames
            run()                                             # (and, with SDFHIP_FLAG_TILE_ORDER, the order every stream's launches will use)
            return min(run() for _ in range(5 if burst else 2))

        tried = []
        for frac in (0.0, 0.04, 0.07, 0.10, 0.13, 0.16, 0.20):
            own = T.balanced_owner(costs, world, extra0=frac * total)
            lay = T.BandLayout(H, world, band_rows, owner=own)
            full = lay.rows_per_rank * W * G
            shares = [torch.zeros(T.sparse2_bytes(lay.rows_per_rank, W, G, full), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
            t0 = job(lay, 0, shares, True)
            t1 = max(job(lay, r, shares, False) for r in sorted({1, world - 1}))
            tried.append((max(t0, t1), frac, t0, t1, own))
            del shares
            if t0 <= t1 and frac > 0:                     # rank 0 is no longer the slowest: charging it more only loads the peers
                break
        _, frac, t0, t1, own = min(tried, key=lambda e: e[0])
        owner = torch.tensor(own, dtype=torch.uint8)
        print("[bench] band deal by cost (" + ("%d-step burst" % steps if burst else "steady state") + "): " +
              ", ".join(f"assembly charged {f:.2f}: rank0 {a * 1e3:.4f} / peers {b * 1e3:.4f} ms" for _, f, a, b, _ in tried) + f" -> {frac:.2f}",
              file=sys.stderr)
        del frames
        torch.cuda.empty_cache()
    if nccl:
        owner = owner.cuda()
    dist.broadcast(owner, src=0)
    return [int(v) for v in owner.cpu().tolist()]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run and relay
    its JSON line and exit code.  This process has not touched the GPU (no torch import yet) and never will."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py")] + sys.argv[1:]
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in child.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return child.returncode if child.returncode or lines else 1
