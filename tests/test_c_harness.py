"""The ABI from plain C (no Python, no PyTorch in the process): compile tests/c_harness.c
against include/sdfhip.h + libsdfhip.so and run it."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, REPO, assert_frames_identical


def build(tmp_path):
    exe = str(tmp_path / "c_harness")
    libdir = os.path.join(REPO, "sdfbox_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "c_harness.c"), "-o", exe, "-L", libdir, "-lsdfhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_c_caller_links_and_reports_missing_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    out = subprocess.run([exe, os.path.join(GOLDEN, "sphere_d4.asdf"), "64", "64", str(tmp_path / "f.raw")],
                         capture_output=True, text=True, timeout=120)
    assert "nodes 3465 depth 4 consistent 1" in out.stdout
    if not torch.cuda.is_available():
        assert out.returncode == 3 and "upload:" in out.stderr       # loud, no fallback
    else:
        assert out.returncode == 0, out.stderr


@pytest.mark.gpu
def test_c_caller_renders_the_oracle_frame(tmp_path, sb, oracle_mod, scenes):
    exe = build(tmp_path)
    raw = tmp_path / "f.raw"
    out = subprocess.run([exe, os.path.join(GOLDEN, "sphere_d4.asdf"), "200", "120", str(raw)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    frame = np.fromfile(raw, dtype=np.float32).reshape(120, 200, 4)
    cam = sb.Logic(200, 120)
    ref, cnt = oracle_mod.render(scenes["sphere_d4"].Structs, scenes["sphere_d4"].Values, cam.State, 200, 120)
    assert_frames_identical(frame, ref, "C harness")
    assert f"{int(cnt[0])} node reads, {int(cnt[1])} samples, {int(cnt[2])} steps" in out.stdout
    assert "page-locked frame identical" in out.stdout            # sdfhip_host_alloc + the same sdfhip_render, from C


@pytest.mark.gpu
def test_c_host_with_frames_in_flight(tmp_path):
    # tests/c_frames_in_flight.c (the measurement behind INTEGRATION.md section 3's hardware-queue paragraph: scripts/hw_queues_c_host.sh)
    # stays buildable and runnable: plain C + four HIP streams + the C ABI, a small scene, a few frames per pass; and the library's
    # export of GPU_MAX_HW_QUEUES reaches a host that set nothing (the program prints what its environment holds after the load)
    exe = str(tmp_path / "c_frames_in_flight")
    libdir = os.path.join(REPO, "sdfbox_amd")
    subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-Wall", "-I", os.path.join(REPO, "include"), "-I", "/opt/rocm/include",
                           os.path.join(REPO, "tests", "c_frames_in_flight.c"), "-o", exe, "-L", libdir, "-lsdfhip", "-L", "/opt/rocm/lib",
                           "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "SDFHIP_KEEP_ENV")}
    out = subprocess.run([exe, "40", "4", "6"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GPU_MAX_HW_QUEUES=8:" in out.stdout and "40 frames on 4 streams" in out.stdout, out.stdout
    kept = subprocess.run([exe, "40", "4", "6"], capture_output=True, text=True, timeout=300, env=dict(env, SDFHIP_KEEP_ENV="1"))
    assert kept.returncode == 0 and "GPU_MAX_HW_QUEUES=(unset: the runtime's 4)" in kept.stdout, kept.stdout + kept.stderr
