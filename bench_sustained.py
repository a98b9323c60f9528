"""bench_sustained.py -- the bench under CONTINUOUS operation.

The reference renders in an endless loop (Program.cs:58-74 -> Draw, :79-110); bench.py's timed region is a burst (the driver's
20 steps are 1.8 ms of GPU work on a chip that has just been idle).  For a kernel bound by instruction issue the sustained
shader clock IS the result, so the default run ends with legs that render the same frames, pipelined exactly as the timed region
does, for seconds of wall time while a host thread samples the GPU's shader clock (each XCD's), socket power, hotspot
temperature and activity every 100 ms:

    telemetry   GpuTelemetry: amdsmi (amdsmi_get_gpu_metrics_info: 0.34 ms per sample on the box) with the hwmon files under the
                device's PCI node as the fallback (freq1_input = sclk, power1_input, temp2_input = junction; gpu_busy_percent)
    the leg     sustained_leg(): frames go round robin to the run's own streams; the host stays at most two chunks (2 x 256
                frames) ahead of the GPU by waiting on an EVENT of two chunks ago -- the pipeline never drains, nothing is
                synchronised inside the leg -- and every chunk boundary is a HIP event on every stream, so the time of the first 20
                frames, of every second and of the last 1000 frames come from the device's own clock

Nothing here is the headline: `value` stays the driver-verifiable K-step region; the line's `sustained` block says what that
number becomes when the loop does not stop.
"""
import glob
import os
import threading
import time

import numpy as np


class GpuTelemetry:
    """Shader clock / power / temperature / activity of ONE GPU, sampled on a daemon thread every `period` seconds."""

    def __init__(self, pci_bus_id, period=0.1):
        self.bdf = (pci_bus_id or "").lower()
        self.period = period
        self.samples = []                 # (t, sclk_mhz_mean, sclk_mhz_min, sclk_mhz_max, power_w, temp_c, activity_pct)
        self.source = None
        self.error = None
        self._stop = threading.Event()
        self._thread = None
        self._smi = self._handle = None
        self._hwmon = self._busy = None
        self._open()

    # -- sources -------------------------------------------------------------------------------------------------------
    def _open(self):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            handles = amdsmi.amdsmi_get_processor_handles()
            pick = None
            for h in handles:
                try:
                    if amdsmi.amdsmi_get_gpu_device_bdf(h).lower() == self.bdf:
                        pick = h
                except Exception:
                    pass
            if pick is None and len(handles) == 1:
                pick = handles[0]
            if pick is not None:
                amdsmi.amdsmi_get_gpu_metrics_info(pick)          # (raises where the driver does not answer)
                self._smi, self._handle, self.source = amdsmi, pick, "amdsmi_get_gpu_metrics_info"
                return
            self.error = f"amdsmi: no handle with PCI bus id {self.bdf} among {len(handles)}"
        except Exception as e:                                     # not importable, no driver, no permission: the files below
            self.error = f"amdsmi: {type(e).__name__}: {str(e).strip()[:120]}"
        for node in ([f"/sys/bus/pci/devices/{self.bdf}"] if self.bdf else []) + sorted(glob.glob("/sys/class/drm/card*/device")):
            mons = sorted(glob.glob(os.path.join(node, "hwmon", "hwmon*")))
            if mons and os.path.exists(os.path.join(mons[0], "freq1_input")):
                self._hwmon, self._busy = mons[0], os.path.join(node, "gpu_busy_percent")
                self.source = f"sysfs {mons[0]} (freq1_input, power1_input, temp2_input)"
                return
        self.error = (self.error or "") + "; no hwmon node with freq1_input under the device's PCI node"

    @staticmethod
    def _num(v):
        return float(v) if isinstance(v, (int, float)) and not isinstance(v, bool) else None

    def _read(self):
        if self._smi is not None:
            m = self._smi.amdsmi_get_gpu_metrics_info(self._handle)
            clks = [float(c) for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
            if not clks and self._num(m.get("current_gfxclk")):
                clks = [float(m["current_gfxclk"])]
            return (float(np.mean(clks)) if clks else None, min(clks) if clks else None, max(clks) if clks else None,
                    self._num(m.get("current_socket_power")) or self._num(m.get("average_socket_power")),
                    self._num(m.get("temperature_hotspot")), self._num(m.get("average_gfx_activity")))

        def rd(path, scale):
            try:
                with open(path) as f:
                    return float(f.read().split()[0]) / scale
            except (OSError, ValueError, IndexError):
                return None
        clk = rd(os.path.join(self._hwmon, "freq1_input"), 1e6)
        return (clk, clk, clk, rd(os.path.join(self._hwmon, "power1_input"), 1e6) or rd(os.path.join(self._hwmon, "power1_average"), 1e6),
                rd(os.path.join(self._hwmon, "temp2_input"), 1e3), rd(self._busy, 1.0))

    # -- sampling ------------------------------------------------------------------------------------------------------
    def _run(self):
        nxt = time.monotonic()
        while not self._stop.is_set():
            try:
                self.samples.append((time.monotonic(),) + tuple(self._read()))
            except Exception as e:
                self.error = f"{type(e).__name__}: {str(e).strip()[:120]}"
            nxt += self.period
            self._stop.wait(max(0.0, nxt - time.monotonic()))

    def start(self):
        self.samples = []
        self._stop.clear()
        if self.source is None:
            return self
        self._thread = threading.Thread(target=self._run, name="bench-telemetry", daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
            self._thread = None
        return self.summary()

    def idle_sample(self):
        """one reading now (before a leg: what the chip looks like at rest)"""
        if self.source is None:
            return None
        try:
            c, lo, hi, p, t, a = self._read()
            return {"sclk_mhz": c, "power_w": p, "temp_c": t, "activity_pct": a}
        except Exception:
            return None

    def summary(self, skip_seconds=0.3):
        """min / mean / max over the samples, the first `skip_seconds` (the ramp out of the idle clock) reported separately"""
        if self.source is None:
            return {"source": None, "error": self.error, "samples": 0}
        if not self.samples:
            return {"source": self.source, "error": self.error, "samples": 0}
        t0 = self.samples[0][0]
        body = [s for s in self.samples if s[0] - t0 >= skip_seconds] or self.samples

        def col(rows, i):
            return [r[i] for r in rows if r[i] is not None]

        def mmm(vals, nd=1):
            return None if not vals else {"min": round(min(vals), nd), "mean": round(float(np.mean(vals)), nd), "max": round(max(vals), nd)}
        out = {"source": self.source, "samples": len(self.samples), "period_s": self.period,
               "sclk_mhz": mmm(col(body, 1)),
               # the slowest and the fastest XCD of a sample (amdsmi reports eight clocks): how far apart the dies run
               "sclk_mhz_slowest_xcd_min": (round(min(col(body, 2)), 1) if col(body, 2) else None),
               "sclk_mhz_fastest_xcd_max": (round(max(col(body, 3)), 1) if col(body, 3) else None),
               "power_w": mmm(col(body, 4)), "temp_c": mmm(col(body, 5)), "activity_pct": mmm(col(body, 6), 0),
               "first_sample": {"sclk_mhz": self.samples[0][1], "power_w": self.samples[0][4], "temp_c": self.samples[0][5]},
               "last_sample": {"sclk_mhz": self.samples[-1][1], "power_w": self.samples[-1][4], "temp_c": self.samples[-1][5]}}
        if self.error:
            out["error"] = self.error
        return out


def sustained_leg(torch, launch, streams, seconds, telemetry, chunk=256, max_frames=4_000_000):
    """Render frames for at least `seconds` of wall time without ever draining the pipeline.

    launch(k, stream_index) issues frame k on streams[stream_index] (frame k goes to stream k % len(streams), as in the timed
    region).  Every `chunk` frames an event is recorded on every stream; the host waits for the events of two chunks ago before
    it issues the next chunk (so it runs at most 2 chunks ahead and the GPU always has >= 1 chunk queued).  Returns the device
    times: whole leg, first 20 frames, last 1000 frames, per second."""
    ns = len(streams)
    torch.cuda.synchronize()
    idle = telemetry.idle_sample() if telemetry is not None else None
    if telemetry is not None:
        telemetry.start()
    start_ev = []
    for s in streams:                                             # the leg's zero on every stream
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        start_ev.append(e)
    marks = []                                                    # (frames issued so far, [event per stream])
    t0 = time.perf_counter()
    k = 0
    first20 = None

    def mark():
        evs = []
        for s in streams:
            e = torch.cuda.Event(enable_timing=True)
            e.record(s)
            evs.append(e)
        marks.append((k, evs))

    while True:
        n = chunk if k else 20                                    # the first mark after 20 frames: the driver's burst, in this leg
        for _ in range(n):
            launch(k, k % ns)
            k += 1
        mark()
        if len(marks) >= 3:
            for e in marks[-3][1]:
                e.synchronize()
        if (time.perf_counter() - t0 >= seconds and len(marks) >= 8) or k >= max_frames:
            break
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tele = telemetry.stop() if telemetry is not None else None

    # Device times.  The streams are not in lock step (each hardware queue runs its own frames; one may lag the others by a good part
    # of a chunk), so "when had ALL streams passed mark m" is late by that lag at every mark but the last -- a window that ends at the
    # leg's end would look short, one that starts at its beginning long.  Windows are therefore timed PER STREAM (a stream's events
    # bracket exactly its own frames: frame k goes to stream k % ns) and the streams' rates added.
    T = [[z.elapsed_time(e) for z, e in zip(start_ev, evs)] for _, evs in marks]      # T[m][s]: ms since stream s's zero at mark m
    K = [kk for kk, _ in marks]

    def own(s_idx, k0, k1):                                       # frames of stream s_idx among frames k0 .. k1 - 1
        return (k1 - s_idx + ns - 1) // ns - (k0 - s_idx + ns - 1) // ns

    def window(m0, m1):                                           # ms per frame between marks m0 (-1: the leg's zero) and m1
        rate = 0.0
        for s_idx in range(ns):
            t0_, k0 = (0.0, 0) if m0 < 0 else (T[m0][s_idx], K[m0])
            dt, n = T[m1][s_idx] - t0_, own(s_idx, k0, K[m1])
            if n > 0 and dt > 0:
                rate += n / dt
        return 1.0 / rate if rate > 0 else float("nan")
    total_ms = max(T[-1])                                         # the leg ends when its last frame does, whichever stream holds it
    first20 = window(-1, 0)
    # the last >= 1000 frames BEFORE the drain: in the leg's final two chunks the streams that are ahead (they keep a lead of 5-12 ms
    # over the slowest one) finish early and the others speed up -- an end effect of 1-2 % of the last 1000 frames, not the steady state
    end = max(1, len(marks) - 3)
    j = end
    while j > 0 and K[end] - K[j] < 1000:
        j -= 1
    last = window(j, end)
    times = [(kk, max(row)) for kk, row in zip(K, T)]
    per_second, edge, prev = [], 1000.0, -1
    for m, (kk, t) in enumerate(times):
        if t >= edge:
            per_second.append(round(window(prev, m), 4))
            prev, edge = m, edge + 1000.0
    return {"frames": k, "seconds": round(total_ms / 1e3, 3), "wall_seconds": round(wall, 3),
            "ms_per_step": round(total_ms / k, 4),
            "ms_per_step_first_20": round(first20, 4),
            "ms_per_step_last_1000": round(last, 4), "last_frames": K[end] - K[j],
            "last_frames_are": "the last >= 1000 frames before the leg's final two chunks (the drain: the streams that are ahead finish early)",
            "windows_are": "per stream (a stream's events bracket its own frames), the streams' rates added: the streams are not in lock step",
            "ms_per_step_each_second": per_second,
            "chunk_frames": chunk, "host_runs_ahead_by_at_most_chunks": 2,
            "gpu_before": idle, "telemetry": tele}


def at_observed_clock(valu_insts_per_frame, ms_per_step, sclk_mhz_mean):
    """valu_frac_of_spec with the spec's 2.4 GHz replaced by the clock the leg ran at: issued VALU wave instructions per second over
    256 CUs x 4 SIMDs x sclk / 2 cycles per wave64 instruction"""
    if not valu_insts_per_frame or not sclk_mhz_mean:
        return None
    peak = 256 * 4 * (sclk_mhz_mean * 1e6) / 2.0
    return round(valu_insts_per_frame / (ms_per_step * 1e-3) / peak, 4)


def sustained_seconds(spec, headline):
    """--sustained -> (cfg2, cfg3, orbit) seconds; 'auto' = 5, 3, 2 for the headline's command, nothing otherwise"""
    spec = (spec or "auto").strip().lower()
    if spec == "auto":
        return (5.0, 3.0, 2.0) if headline else (0.0, 0.0, 0.0)
    if spec in ("0", "off", "none", ""):
        return (0.0, 0.0, 0.0)
    v = [float(x) for x in spec.split(",")]
    return tuple((v + [0.0, 0.0, 0.0])[:3])
