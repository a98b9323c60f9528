"""bench.py prints one JSON line with the contract's keys (small workload), starts its own ranks for
--gpus N, and its sharded pipeline -- bands, wire shares, gather, assembly, dense resend -- reproduces the
whole-frame render at BASELINE cfg-4's frame size."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "latency_ms", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*args, timeout=360):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + [str(a) for a in args],
                         capture_output=True, text=True, timeout=timeout, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_small_workload():
    j = run_bench("--steps", 3, "--warmup", 1, "--depth", 6, "--size", "320x200", "--cpu-seconds", 0.5)
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["unit"] == "Mray/s" and j["dtype"] == "f32"
    assert j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    # no PMC pass exists for this toy workload: no fraction is reported, only what the kernel asks of the memory system
    assert r["pmc_stale"] is True and r["frac"] is None and r["traffic"] is None and r["bound"] is None
    assert r["demand"]["own_bytes_per_frame"] > 0 and r["demand"]["reference_bytes_per_frame"] > r["demand"]["own_bytes_per_frame"]
    assert r["demand"]["own_demand_over_hbm_peak"] > 0 and "configs" not in j          # (not the headline's command line)
    assert j["latency"]["ms"] == j["latency_ms"] > 0 and j["latency"]["orbit_ms"] > 0
    assert j["latency_ms_tile_order"] == j["latency"]["tile_order"]["ms"] > 0 and j["latency_ms_tile_order_moving_camera"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mray/s" and c["value"] > 0
    # the thread count is swept; `cores` = the threads of the best run, never more than the affinity mask holds
    assert 1 <= c["cores"] <= c["limits"]["affinity_cpus"] and c["value"] == max(e["value"] for e in c["sweep"])
    assert [e["threads"] for e in c["sweep"]][0] == 1 and c["one_thread"]["value"] == c["sweep"][0]["value"]
    assert c["native_build_equals_portable_build"] in (True, None)


@pytest.mark.gpu
def test_bench_gpus_n_starts_its_own_ranks():
    # `python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py spawns torch.distributed.run itself
    # (here both ranks share the one GPU through gloo) and relays the one JSON line
    j = run_bench("--gpus", 2, "--backend", "gloo", "--steps", 8, "--warmup", 4, "--depth", 6, "--size", "640x360", "--check")
    assert j["n_gpus"] == 2 and j["config"]["assembled_frame_equals_whole_frame_render"] is True
    assert j["config"]["gather_format"] == "sparse shares written by the march kernel"
    # the display pass travels as the frame's own RGBA8 pixels
    j = run_bench("--gpus", 2, "--backend", "gloo", "--steps", 8, "--warmup", 4, "--depth", 6, "--size", "640x360", "--check", "--display")
    assert j["config"]["assembled_frame_equals_whole_frame_render"] is True and j["config"]["gather_format"] == "frame pixels"


@pytest.mark.gpu
def test_cfg4_frame_through_the_sharded_pipeline_one_rank():
    # BASELINE cfg-4's frame (3840x2160) through the NCCL code path with one rank: band render into wire shares,
    # sparse compaction, gather, assembly; every assembled frame must equal the whole-frame render bit for bit
    j = run_bench("--exercise-gather", "--check", "--size", "3840x2160", "--steps", 8, "--warmup", 4, "--no-cpu-baseline")
    assert j["config"]["assembled_frame_equals_whole_frame_render"] is True
    assert j["config"]["gather_format"] == "sparse shares written by the march kernel" and j["config"]["float_tails_sent_again"] == 0


@pytest.mark.gpu
def test_cfg4_frame_two_ranks_moving_camera_and_dense_resend():
    # two ranks (gloo, one GPU) on the 4K frame with a camera that moves every frame and far too few floats travelling with
    # the shares: the tails must come again, point to point, and the assembled frames must still be exact
    j = run_bench("--gpus", 2, "--backend", "gloo", "--check", "--size", "3840x2160", "--steps", 8, "--warmup", 4,
                  "--orbit", 16, "--sparse-cap-scale", 0.2, "--no-cpu-baseline")
    assert j["n_gpus"] == 2 and j["config"]["assembled_frame_equals_whole_frame_render"] is True
    assert j["config"]["float_tails_sent_again"] > 0


@pytest.mark.gpu
def test_partial_last_group_renders_its_own_frames_only():
    # the driver's scaling run times 20 steps; at 8 ranks a group is 8 frames, so the last group holds 4: it must render (and
    # expand) those 4 only -- the padded form rendered 24 frames for 20 steps -- and the frames must still be exact, resends included
    j = run_bench("--exercise-gather", "--check", "--gather-every", 8, "--steps", 20, "--warmup", 5, "--depth", 6, "--size", "640x360",
                  "--no-cpu-baseline")
    assert j["config"]["frames_per_gather"] == 8 and j["config"]["frames_rendered_in_the_timed_region"] == 20
    assert j["config"]["assembled_frame_equals_whole_frame_render"] is True
    j = run_bench("--gpus", 2, "--backend", "gloo", "--check", "--steps", 10, "--warmup", 3, "--depth", 6, "--size", "640x360",
                  "--orbit", 16, "--sparse-cap-scale", 0.2, "--no-cpu-baseline")
    assert j["n_gpus"] == 2 and j["config"]["frames_rendered_in_the_timed_region"] == 10
    assert j["config"]["assembled_frame_equals_whole_frame_render"] is True and j["config"]["float_tails_sent_again"] > 0


@pytest.mark.gpu
def test_cfg5_frame_through_the_sharded_pipelines():
    # BASELINE cfg-5's frame (3840x2160, 16 spp, 3 bounces) through the gather pipelines: the NCCL code path with one rank, two
    # gloo ranks on the one GPU, and the library's own multi-device entry points over the device list [0, 0, 0, 0]; every
    # assembled frame must equal the whole-frame render bit for bit
    j = run_bench("--exercise-gather", "--check", "--size", "3840x2160", "--spp", 16, "--steps", 2, "--warmup", 1, "--no-cpu-baseline")
    assert j["config"]["assembled_frame_equals_whole_frame_render"] is True and j["config"]["gather_format"] == "frame pixels"
    j = run_bench("--gpus", 2, "--backend", "gloo", "--check", "--size", "3840x2160", "--spp", 16, "--steps", 2, "--warmup", 1, "--no-cpu-baseline")
    assert j["n_gpus"] == 2 and j["config"]["assembled_frame_equals_whole_frame_render"] is True
    j = run_bench("--single-process", "--devices", "0,0,0,0", "--check", "--size", "3840x2160", "--spp", 16, "--steps", 2, "--warmup", 1)
    assert j["n_gpus"] == 4 and j["config"]["assembled_frame_equals_whole_frame_render"] is True


@pytest.mark.gpu
def test_single_process_bench_line():
    # `bench.py --single-process`: the N-device frame through sdfhip_multi_submit / _wait, groups and one frame at a time
    for mode in ("groups", "frame"):
        j = run_bench("--single-process", "--devices", "0,0,0", "--check", "--size", "1280x720", "--depth", 7, "--steps", 24, "--warmup", 8,
                      "--multi-mode", mode, "--orbit", 8)
        for k in REQUIRED:
            if k != "cpu_baseline":
                assert k in j, k
        assert j["n_gpus"] == 3 and j["config"]["assembled_frame_equals_whole_frame_render"] is True
        assert ("one frame at a time" in j["config"]["measures"]) == (mode == "frame")


@pytest.mark.gpu
def test_bench_line_carries_the_other_single_gpu_configs():
    # `configs`: cfg-3 (compaction off / on), cfg-5 and the deeper scene timed in the same process (scaled down here)
    j = run_bench("--steps", 3, "--warmup", 1, "--depth", 6, "--size", "320x200", "--no-cpu-baseline", "--configs", "all", "--configs-scale", 8)
    c = j["configs"]
    assert set(c) == {"cfg3_4k", "cfg3_4k_compact", "cfg5_4k_spp16", "cfg2_depth10"}
    for name, e in c.items():
        assert e["ms_per_step"] > 0 and e["value"] > 0 and e["unit"] == "Mray/s" and e["steps"] > 0 and e["kernel_ms"] > 0, name
        assert e["pmc_stale"] is True and e["hbm_frac"] is None and e["valu_frac_of_spec"] is None and len(e["kernel_source_sha"]) == 16, name
    assert "480x270" in c["cfg5_4k_spp16"]["workload"] and "16 spp" in c["cfg5_4k_spp16"]["workload"] and "dragon_standin_d7" in c["cfg2_depth10"]["workload"]


def test_committed_pmc_passes_belong_to_this_build():
    # The bench line's fractions come from profiles/hbm_traffic.json, which is only valid for the kernel sources it was measured
    # on.  At commit time the headline's entry -- and the entries of the `configs` block -- must carry the hash of the sources in
    # the tree: a kernel edit without a re-profile (scripts/profile_all.sh + scripts/summarise_all.py) fails here, not silently
    # in the driver's bench line.
    sys.path.insert(0, REPO)
    import bench
    with open(os.path.join(REPO, "profiles", "hbm_traffic.json")) as f:
        t = json.load(f)
    here = bench.kernel_source_hash()
    for key in ("1920x1080:dragon_standin_d9:default:grid8+blocks", "3840x2160:dragon_standin_d9:default:grid8+blocks",
                "3840x2160:dragon_standin_d9:compact:grid8+blocks", "3840x2160:dragon_standin_d9:spp16:grid8",
                "1920x1080:dragon_standin_d10:default:grid8+blocks", "1920x1080:knot_d10.asdf:default:grid7+blocks"):
        assert key in t, key
        assert t[key]["kernel_source_sha"] == here, f"{key}: measured on {t[key]['kernel_source_sha']}, the tree is {here}: re-profile"
        assert "dropped" not in bench.load_pmc(key)


@pytest.mark.gpu
def test_bench_ab_forms_need_the_experiments_flavour():
    # --one-kernel / --shadow-queue are A/B forms of libsdfhip_lab.so: `--lab` loads it; without it bench.py says so
    j = run_bench("--lab", "--one-kernel", "--steps", 3, "--warmup", 1, "--depth", 6, "--size", "320x200", "--no-cpu-baseline")
    assert "one kernel" in j["config"]["kernel"]
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--shadow-queue", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0 and "--lab" in (out.stderr + out.stdout)


def test_roofline_reports_only_measured_fractions():
    # bench.py's roofline(): `frac` is the fraction SURVEY.md 8d asks for -- HBM counter bytes / time / 8 TB/s (bound "hbm") --
    # with the issued-VALU fraction of the chip's spec rate as its PEER and `limiting` naming the larger; fractions of the
    # MEASURED streaming rate beside the nameplate ones; without a PMC entry of this build no fraction at all, only the demand
    # figures, which are labelled as not being fractions
    sys.path.insert(0, REPO)
    import bench
    assert bench.VALU_PEAK_SPEC_GINSTR == 1228.8
    pmc = {"hbm_bytes_per_frame": 580_000_000, "valu_insts_per_frame": 68_000_000, "profile": "profiles/x.json", "kernel_source_sha": "abc"}
    measured = {"copy_gbs": 6100.0, "triad_gbs": 6300.0, "best_gbs": 6300.0}
    r = bench.roofline(0.1232e-3, 904e6, 3.92e9, pmc, measured)
    assert r["bound"] == "hbm" and r["limiting"] == "hbm" and r["traffic"] == 580_000_000 and r["pmc_stale"] is False and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - 580e6 / 0.1232e-3 / 1e9 / 8000.0) < 1e-3 and r["frac"] == r["hbm_frac"] <= 1.0
    assert abs(r["hbm_frac_of_measured"] - 580e6 / 0.1232e-3 / 1e9 / 6300.0) < 1e-3 and r["measured_hbm_gbs"] == measured
    assert abs(r["candidates"]["valu"]["frac"] - 68e6 / 0.1232e-3 / 1e9 / 1228.8) < 1e-3 and r["valu_frac_of_spec"] == r["candidates"]["valu"]["frac"]
    assert abs(r["valu_frac_of_measured_ceiling"] - 68e6 / 0.1232e-3 / 1e9 / bench.VALU_PEAK_GINSTR) < 1e-3
    assert r["demand"]["own_bytes_per_frame"] == 904_000_000 and "algorithmic" not in r
    # an issue-bound frame: `frac` stays the HBM fraction (what 8d asks), `limiting` says valu and the peer carries its fraction
    r4k = bench.roofline(0.388e-3, 3.62e9, 15.7e9, {"hbm_bytes_per_frame": 1_178_000_000, "valu_insts_per_frame": 259_000_000}, None)
    assert r4k["bound"] == "hbm" and r4k["limiting"] == "valu" and r4k["frac"] == r4k["hbm_frac"] < r4k["valu_frac_of_spec"] <= 1.0
    assert r4k["hbm_frac_of_measured"] is None and r4k["candidates"]["valu"]["peak"] == 1228.8
    none = bench.roofline(0.388e-3, 3.62e9, 15.7e9, None, None)      # the demand of 64 lanes asking for the same cells is not an HBM figure
    assert none["bound"] is None and none["limiting"] is None and none["traffic"] is None and none["pmc_stale"] is True and none["frac"] is None
    assert none["demand"]["own_demand_over_hbm_peak"] > 1.0 and none["demand"]["reference_demand_over_hbm_peak"] > 5.0     # 9.3 TB/s of demand is no HBM fraction
    # PMC figures are reported only for the build they were measured on
    assert "dropped" in bench.load_pmc("no such workload")            # ... and the line says why there are none
    gone = bench.roofline(0.1e-3, 1.0e9, 4.0e9, bench.load_pmc("no such workload"), 5000.0)
    assert gone["traffic"] is None and "no PMC pass" in gone["traffic_source"] and gone["hbm_frac"] is None and gone["valu_busy"] is None and gone["pmc_stale"]
    full = bench.roofline(0.098e-3, 968e6, 3.92e9, {"hbm_bytes_per_frame": 236_142_336, "valu_insts_per_frame": 59_606_143,
                                                    "valu_active_quad_cycles_per_frame": 59_900_000, "profile": "p", "kernel_source_sha": "s"}, 6100.0)
    assert abs(full["hbm_frac"] - 236_142_336 / 0.098e-3 / 8e12) < 1e-3 and abs(full["hbm_frac_of_measured"] - 236_142_336 / 0.098e-3 / 6.1e12) < 1e-3
    assert abs(full["valu_busy"] - 59.9e6 * 4 / (1024 * 2.4e9 * 0.098e-3)) < 1e-3 and "not a measured utilisation" in full["valu_busy_is"]
    assert full["limiting"] == "valu" and abs(full["valu_frac_of_spec"] - 59_606_143 / 0.098e-3 / 1228.8e9) < 1e-3 and full["frac"] == full["hbm_frac"]
    assert full["candidates"]["valu"]["measured_ceiling"] == round(bench.VALU_PEAK_GINSTR, 1) and full["traffic_source"]["frames_in_flight_while_counting"] == 1
    import json as _json
    with open(os.path.join(REPO, "profiles", "hbm_traffic.json")) as f:
        t = _json.load(f)
    for key, e in t.items():
        assert set(e) >= {"hbm_bytes_per_frame", "valu_insts_per_frame", "profile", "kernel_source_sha"}, key
        assert os.path.exists(os.path.join(REPO, e["profile"])), key


def test_configs_summary_is_short_and_names_every_config():
    # the driver keeps the last 2 000 characters of the line: `configs_summary`, its LAST key, carries every configuration's time
    # and both fractions in <= 600 characters
    sys.path.insert(0, REPO)
    import bench
    cfgs = {n: {"ms_per_step": 0.3066 + i, "value": 27012.3, "hbm_frac": 0.1712, "hbm_frac_of_measured": 0.2175, "valu_frac_of_spec": 0.5431,
                "pmc_stale": False} for i, n in enumerate(("cfg3_4k", "cfg3_4k_compact", "cfg5_4k_spp16", "cfg2_depth10", "cfg2_mesh_knot_d10"))}
    cfgs["cfg5_4k_spp16"] = {"error": "RuntimeError: out of memory " + "x" * 300}
    cfgs["cfg2_depth10"].update(pmc_stale=True, hbm_frac=None, valu_frac_of_spec=None, hbm_frac_of_measured=None)
    text = bench.configs_summary({"ms_per_step": 0.0901, "value": 23014.0, "roofline": {"hbm_frac": 0.33, "hbm_frac_of_measured": 0.42, "valu_frac_of_spec": 0.52}}, cfgs)
    assert len(text) <= 600 and text.startswith("ms/Mray/hbm/hbm_meas/valu: cfg2 0.0901/23014/0.33/0.42/0.52")
    for n in ("cfg3 ", "cfg3c ", "cfg5 ", "d10 ", "mesh "):
        assert n in text, n
    assert "cfg5 ERR" in text and "d10 3.3066/27012/-/-/-" in text


def test_sustained_block_reaches_the_summary_and_prices_the_observed_clock():
    # the sustained legs' figures sit in the line's last key (what the driver's record keeps) and
    # valu_frac is recomputed at the clock the leg ran at
    sys.path.insert(0, REPO)
    import bench
    import bench_sustained as bs
    leg = {"ms_per_step": 0.0871, "seconds": 5.02, "frames": 57632, "ms_per_step_first_20": 0.0932, "ms_per_step_last_1000": 0.0869,
           "telemetry": {"sclk_mhz": {"min": 2085.0, "mean": 2240.5, "max": 2400.0}, "power_w": {"min": 900.0, "mean": 1011.0, "max": 1100.0}},
           "valu_frac_at_observed_clock": bs.at_observed_clock(57_043_052, 0.0871, 2240.5)}
    assert abs(leg["valu_frac_at_observed_clock"] - 57_043_052 / 0.0871e-3 / (1024 * 2240.5e6 / 2)) < 1e-4
    assert bs.at_observed_clock(None, 0.1, 2000.0) is None and bs.at_observed_clock(1e6, 0.1, None) is None
    cfgs = {"cfg3_4k": {"ms_per_step": 0.31, "value": 26775.0, "hbm_frac": 0.17, "hbm_frac_of_measured": 0.22, "valu_frac_of_spec": 0.53,
                        "sustained": dict(leg, ms_per_step=0.3099, frames=9876, seconds=3.06)}}
    text = bench.configs_summary({"ms_per_step": 0.0896, "value": 23135.0, "roofline": {"hbm_frac": 0.33, "hbm_frac_of_measured": 0.42,
                                  "valu_frac_of_spec": 0.52}, "sustained": {"cfg2": leg, "cfg2_orbit": dict(leg, ms_per_step=0.0902)}}, cfgs)
    assert len(text) <= 980
    assert "| sustained: cfg2 0.0871ms/5.0s/57632f first20 0.0932 last1000 0.0869 sclk 2085-2240MHz 1011W valu@clk 0.57" in text
    assert "orbit 0.0902ms" in text and "cfg3 0.3099ms/3.1s/9876f" in text
    # ... and so does what the run's own counter passes found (bench_live_pmc.py), or why there is none
    live = {"used": "live", "seconds": 7.7, "hbm_bytes_per_frame": 216319390, "valu_insts_per_frame": 55339941, "live_over_committed": 1.0002}
    text = bench.configs_summary({"ms_per_step": 0.0896, "value": 23135.0, "roofline": {"hbm_frac": 0.33, "live_pmc": live}}, {})
    assert text.endswith("| live pmc (this run, 8s): cfg2 216.3MB 55.34M valu = 1.0002 x committed bytes")
    text = bench.configs_summary({"ms_per_step": 0.0896, "value": 23135.0, "roofline": {"live_pmc": dict(live, orbit={"used": "live", "live_over_committed": 0.9991})}},
                                 {"cfg3_4k": {"ms_per_step": 0.3, "value": 1.0, "live_pmc": {"used": "live", "live_over_committed": 1.0011}},
                                  "cfg2_depth10": {"ms_per_step": 0.1, "value": 1.0, "live_pmc": {"used": "committed", "dropped": "x"}}})
    assert text.endswith("x committed bytes; orbit 0.9991; cfg3 1.0011; d10 dropped")
    text = bench.configs_summary({"ms_per_step": 0.0896, "value": 23135.0, "roofline": {"live_pmc": {"used": "committed", "dropped": "rocprofv3 is not on PATH"}}}, {})
    assert text.endswith("| live pmc dropped: rocprofv3 is not on PATH")
    assert bs.sustained_seconds("auto", True) == (5.0, 3.0, 2.0) and bs.sustained_seconds("auto", False) == (0.0, 0.0, 0.0)
    assert bs.sustained_seconds("off", True) == (0.0, 0.0, 0.0) and bs.sustained_seconds("1.5,0.5", True) == (1.5, 0.5, 0.0)


def test_telemetry_summary_and_the_leg_bookkeeping_without_a_gpu(monkeypatch):
    # GpuTelemetry on a machine without a GPU names why it has no source and summarises nothing; with samples put in, the summary
    # leaves the ramp out of the idle clock (the first 0.3 s) out of min / mean / max.  sustained_leg's bookkeeping (first 20,
    # last >= 1000 frames, per second) is exercised against a stand-in for torch whose events carry a simulated device clock.
    sys.path.insert(0, REPO)
    import bench_sustained as bs
    t = bs.GpuTelemetry("0000:ff:00.0")
    if t.source is None:
        assert t.error and t.start().stop()["samples"] == 0
    t.source = "test"
    t.samples = [(10.0, 95.0, 94.0, 97.0, 245.0, 46.0, 0.0), (10.1, 1200.0, 1100.0, 1300.0, 600.0, 50.0, 80.0)] + \
                [(10.4 + 0.1 * i, 2100.0 + i, 2090.0, 2110.0 + i, 1000.0, 60.0 + i, 100.0) for i in range(10)]
    sm = t.summary()
    assert sm["samples"] == 12 and sm["sclk_mhz"] == {"min": 2100.0, "mean": 2104.5, "max": 2109.0} and sm["first_sample"]["sclk_mhz"] == 95.0
    assert sm["sclk_mhz_slowest_xcd_min"] == 2090.0 and sm["temp_c"]["max"] == 69.0 and sm["power_w"]["mean"] == 1000.0

    class Clock:                       # the simulated device: every launch takes 0.1 ms, four streams in parallel
        now = 0.0
    class Ev:
        def __init__(self, enable_timing=True): self.t = None
        def record(self, s): self.t = s.busy_until
        def synchronize(self): pass
        def elapsed_time(self, other): return other.t - self.t
    class Stream:
        def __init__(self): self.busy_until = 0.0
    class Cuda:
        Event = Ev
        @staticmethod
        def synchronize(): pass
    class Torch:
        cuda = Cuda
    streams = [Stream() for _ in range(4)]

    def launch(k, si):
        streams[si].busy_until += 0.4 if k >= 20 else 0.8          # the first 20 frames twice as slow (a cold chip)
    ticks = iter(range(10 ** 9))
    monkeypatch.setattr(bs.time, "perf_counter", lambda: next(ticks) * 0.05)     # the host's clock: 0.05 s per look
    r = bs.sustained_leg(Torch, launch, streams, 1.0, None, chunk=256)
    assert r["frames"] >= 20 + 7 * 256 and r["frames"] % 256 == 20
    assert abs(r["ms_per_step_first_20"] - 0.2) < 1e-9 and abs(r["ms_per_step_last_1000"] - 0.1) < 1e-9 and r["last_frames"] >= 1000
    assert abs(r["ms_per_step"] - (20 * 0.2 + (r["frames"] - 20) * 0.1) / r["frames"]) < 1e-4 and r["telemetry"] is None
    # streams that are NOT in lock step (one hardware queue slower than the others): windows are timed per stream and the rates added
    streams2 = [Stream() for _ in range(4)]

    def launch2(k, si):
        streams2[si].busy_until += 0.5 if si == 3 else 0.4
    r2 = bs.sustained_leg(Torch, launch2, streams2, 1.0, None, chunk=256)
    want = 1.0 / (3 / 0.4 + 1 / 0.5)
    assert abs(r2["ms_per_step_last_1000"] - want) < 1e-4 and abs(r2["ms_per_step_first_20"] - want) < 1e-4
    assert all(abs(v - want) < 2e-4 for v in r2["ms_per_step_each_second"])
    assert abs(r2["seconds"] * 1e3 - 0.5 * ((r2["frames"] + 0) // 4)) < 1.0          # the leg ends with its slowest stream


def test_orbit_cameras_walk_round_the_scene():
    sys.path.insert(0, REPO)
    import bench
    import sdfbox_amd as sb
    cams = bench.orbit_cameras(sb, 640, 360, 90)
    assert len(cams) == 90
    p0 = list(cams[0].State.position)
    assert [round(v, 6) for v in p0] == [0.5, 0.5, -0.35]          # camera 0 is the cfg-2 camera
    for c in cams:                                                  # all on the circle of radius 0.85 about the cube centre, at its height
        x, y, z = c.State.position
        assert abs(((x - 0.5) ** 2 + (z - 0.5) ** 2) ** 0.5 - 0.85) < 1e-5 and abs(y - 0.5) < 1e-6


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode != 0 and "no CPU path" in (out.stderr + out.stdout)


def test_only_the_cpu_baseline_leg_of_bench_py_touches_the_oracle():
    # the oracle is test infrastructure: outside tests/ and smoke() only bench.py's cpu_baseline leg may import it -- also after
    # bench.py became four modules (the leg stayed in bench.py itself)
    import re
    for f in ("bench_report.py", "bench_configs.py", "bench_sharded.py", "bench_sustained.py", "bench_live_pmc.py"):
        text = open(os.path.join(REPO, f)).read()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
    text = open(os.path.join(REPO, "bench.py")).read()
    hits = [m.start() for m in re.finditer(r"^\s*(import|from)\s+oracle\b", text, re.M)]
    assert len(hits) == 1 and text.rfind("def cpu_baseline(", 0, hits[0]) > text.rfind("\ndef main(", 0, hits[0])


def _fake_counter_csv(path, counter_values, frames=6):
    """a counter_collection.csv as rocprofv3 writes it (the columns bench_live_pmc reads): one counting launch (left out), `frames`
    launches of the default kernel, and a kernel that is not the renderer's"""
    import csv
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "VGPR_Count", "Counter_Name", "Counter_Value"])
        d = 0
        for c, v in counter_values.items():
            d += 1
            w.writerow([d, "void sdfhip::k_march<3, true, 0, false>(sdfhip::RenderParams)", 64, c, v * 50])       # COUNT = true
            for _ in range(frames):
                d += 1
                w.writerow([d, "void sdfhip::k_march<3, false, 0, false>(sdfhip::RenderParams)", 64, c, v])
            w.writerow([d + 1, "void at::native::vectorized_elementwise_kernel<4>(int)", 8, c, 12345.0])


def test_live_pmc_passes_are_parsed_like_the_committed_ones_and_fall_back_loudly(tmp_path):
    # bench_live_pmc.py (the roofline's numerator in the driver's own run) without a GPU: the passes' command
    # lines, the per-frame arithmetic on counter files of rocprofv3's shape, the record handed to roofline(), every way out
    import subprocess as sp
    sys.path.insert(0, REPO)
    import bench_live_pmc as L
    from bench_report import roofline
    cmd = L.child_command([], ("FETCH_SIZE",), "/tmp/x")
    assert cmd[0] == "rocprofv3" and cmd[1:3] == ["--pmc", "FETCH_SIZE"]
    assert cmd[cmd.index("--") + 1] == sys.executable and cmd[cmd.index("--") + 2].endswith("bench.py")     # the program itself after --
    assert not any(a in cmd for a in ("--sys-trace", "-s", "--runtime-trace", "-r", "--kernel-trace", "--stats", "--hip-trace", "--hsa-trace"))
    assert "--only-timed" in cmd and cmd[cmd.index("--live-pmc") + 1] == "off" and cmd[cmd.index("--sustained") + 1] == "off"
    assert all(len(set(p) & {"FETCH_SIZE", "WRITE_SIZE"}) <= 1 for p in L.PASSES)                         # the byte counters: separate passes
    vals = {"FETCH_SIZE": 100000.0, "WRITE_SIZE": 32000.0, "SQ_INSTS_VALU": 56.0e6, "SQ_INSTS_SALU": 20.0e6, "SQ_ACTIVE_INST_VALU": 57.0e6}
    calls = []

    def run(cmd, cwd, env, timeout):
        counters = cmd[cmd.index("--pmc") + 1: cmd.index("--output-format")]
        calls.append(counters)
        assert env.get("GPU_MAX_HW_QUEUES") and env.get("TMPDIR") == "/tmp" and cwd == REPO
        _fake_counter_csv(os.path.join(cmd[cmd.index("-d") + 1], "host", "123_counter_collection.csv"), {c: vals[c] for c in counters})
        return 0, ""
    rec = L.live_pmc([], "abc", run=run, which=lambda _: "/opt/rocm/bin/rocprofv3")
    assert calls == [list(p) for p in L.PASSES]
    assert rec["read_x2"] == int(2 * 100000.0 * 1024) and rec["write"] == int(32000.0 * 1024)               # KB -> bytes, FETCH doubled
    assert rec["hbm_bytes_per_frame"] == rec["read_x2"] + rec["write"] and rec["valu_insts_per_frame"] == 56000000
    assert rec["kernel_source_sha"] == "abc" and all(p["frames_counted"] == 6 for p in rec["live"]["passes"])
    committed = {"hbm_bytes_per_frame": int(rec["hbm_bytes_per_frame"] * 1.02), "valu_insts_per_frame": 57000000, "profile": "profiles/x_pmc.json",
                 "kernel_source_sha": "abc", "valu_active_quad_cycles_per_frame": 1}
    use, note = L.merged(rec, committed)
    assert note["used"] == "live" and abs(note["live_over_committed"] - 1 / 1.02) < 1e-3 and note["committed"]["profile"] == "profiles/x_pmc.json"
    roof = roofline(0.09e-3, 1.0, 1.0, use, 6300.0)
    assert roof["traffic"] == rec["hbm_bytes_per_frame"] and roof["traffic_source"]["profile"].startswith("live:") and roof["valu_frac_of_spec"]
    # ... the third pass may fail (the VALU figure then comes from the committed pass of the same build); a byte pass may not
    def run_no_valu(cmd, cwd, env, timeout):
        if "SQ_INSTS_VALU" in cmd:
            return 1, "no such counter"
        return run(cmd, cwd, env, timeout)
    rec2 = L.live_pmc([], "abc", run=run_no_valu, which=lambda _: "x")
    use2, note2 = L.merged(rec2, committed)
    assert "valu_insts_per_frame" not in rec2 and use2["valu_insts_per_frame"] == 57000000 and note2["used"] == "live"
    for bad in ("FETCH_SIZE", "WRITE_SIZE"):
        def run_bad(cmd, cwd, env, timeout, bad=bad):
            return (1, "boom") if bad in cmd else run(cmd, cwd, env, timeout)
        r = L.live_pmc([], "abc", run=run_bad, which=lambda _: "x")
        assert "dropped" in r and bad in r["dropped"]
        use3, note3 = L.merged(r, committed)
        assert use3 is committed and note3["used"] == "committed" and bad in note3["dropped"]

    def run_hangs(cmd, cwd, env, timeout):
        raise sp.TimeoutExpired(cmd, timeout)
    assert "did not end within" in L.live_pmc([], "abc", run=run_hangs, which=lambda _: "x")["dropped"]
    assert "not on PATH" in L.live_pmc([], "abc", run=run, which=lambda _: None)["dropped"]
    # a bench.py that is itself being profiled starts no profiler of its own
    assert L.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so:/opt/rocm/lib/librocprofiler-sdk.so"})
    assert L.under_profiler({"ROCPROFILER_LIBRARY_CTOR": "1"}) and not L.under_profiler({"LD_PRELOAD": "libfoo.so", "PATH": "/opt/rocm/bin"})
    # a pass that ends without a row of the renderer's kernels is no measurement
    def run_empty(cmd, cwd, env, timeout):
        _fake_counter_csv(os.path.join(cmd[cmd.index("-d") + 1], "h", "1_counter_collection.csv"), {})
        return 0, ""
    assert "dropped" in L.live_pmc([], "abc", run=run_empty, which=lambda _: "x")


def test_run_group_ends_the_whole_group_at_its_timeout(tmp_path):
    # the pass's child process group: a child that starts a grandchild and both sleep -- after the timeout neither is left
    import subprocess as sp
    import time
    sys.path.insert(0, REPO)
    import bench_live_pmc as L
    pidfile = tmp_path / "pids"
    code = (f"import os, subprocess, sys, time; p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(60)']); "
            f"open({str(pidfile)!r}, 'w').write(f'{{os.getpid()}} {{p.pid}}'); time.sleep(60)")
    with pytest.raises(sp.TimeoutExpired):
        L.run_group([sys.executable, "-c", code], str(tmp_path), dict(os.environ), 2.0)
    time.sleep(0.3)
    for pid in (int(x) for x in pidfile.read_text().split()):
        try:
            os.kill(pid, 0)
            state = open(f"/proc/{pid}/stat").read().split()[2]
            assert state == "Z", f"process {pid} survived the group's end (state {state})"
        except (ProcessLookupError, FileNotFoundError):
            pass
    assert L.run_group([sys.executable, "-c", "import sys; sys.stderr.write('x' * 1000); sys.exit(3)"], str(tmp_path), dict(os.environ), 30.0) == (3, "x" * 300)


@pytest.mark.gpu
def test_headline_counts_its_own_bytes_through_child_passes():
    # the headline's command with everything behind the timed region switched off but the live counter passes: roofline.traffic comes
    # from rocprofv3 --pmc children of this very run and agrees with the committed pass of the same build
    j = run_bench("--steps", 5, "--warmup", 1, "--no-cpu-baseline", "--configs", "none", "--sustained", "off")
    r = j["roofline"]
    lp = r["live_pmc"]
    if lp["used"] != "live":                                  # a box without a usable rocprofv3: the committed figure, and the reason
        assert lp["dropped"] and r["traffic_source"]["live"] is False and not str(r["traffic_source"]["profile"]).startswith("live")
        pytest.skip("no live pass on this box: " + str(lp["dropped"]))
    assert r["traffic"] == lp["hbm_bytes_per_frame"] and r["traffic_source"]["live"] is True and r["traffic_source"]["profile"].startswith("live:")
    assert [p["counters"] for p in lp["passes"]][:2] == [["FETCH_SIZE"], ["WRITE_SIZE"]] and all(p["exit"] == 0 and p["frames_counted"] >= 5 for p in lp["passes"])
    assert 0.97 < lp["live_over_committed"] < 1.03, lp
    assert lp["valu_insts_per_frame"] == lp["committed"]["valu_insts_per_frame"]          # the instruction count of a frame is deterministic
    assert lp["orbit"]["used"] == "live" and 0.97 < lp["orbit"]["live_over_committed"] < 1.03 and r["orbit"]["traffic"] == lp["orbit"]["hbm_bytes_per_frame"]
    assert "x committed bytes; orbit " in j["configs_summary"]
