"""The C++ host mirror of the reference's interface (include/sdfbox.hpp: OctData, Logic,
Program) above the C ABI: compile tests/cpp_host.cpp and run the reference's own flow
Logic.MakeData -> Program.Load -> Program.Draw."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, REPO, assert_frames_identical


def build(tmp_path):
    exe = str(tmp_path / "cpp_host")
    libdir = os.path.join(REPO, "sdfbox_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "cpp_host.cpp"), "-o", exe, "-L", libdir, "-lsdfhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    shutil.copy(os.path.join(GOLDEN, "sphere_d4.asdf"), tmp_path / "sphere.asdf")
    # the model name is given without its extension: Logic.AutocompleteFile finds sphere.asdf
    out = subprocess.run([exe, str(tmp_path / "sphere"), "64", "64", str(tmp_path / "f.raw")],
                         capture_output=True, text=True, timeout=120)
    assert "Length 3465 buffer_size 3465" in out.stdout
    if torch.cuda.is_available():
        assert out.returncode == 0, out.stderr
    else:
        assert out.returncode == 3 and "sdfhip:" in out.stderr


@pytest.mark.gpu
def test_cpp_host_flow_matches_oracle(tmp_path, sb, oracle_mod, scenes):
    from test_sdfgen import fib_sphere, write_ply
    exe = build(tmp_path)
    shutil.copy(os.path.join(GOLDEN, "sphere_d4.asdf"), tmp_path / "sphere.asdf")
    raw, disp = tmp_path / "f.raw", tmp_path / "d.raw"
    out = subprocess.run([exe, str(tmp_path / "sphere"), "160", "96", str(raw), str(disp)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cam = sb.Logic(160, 96)
    ref, _ = oracle_mod.render(scenes["sphere_d4"].Structs, scenes["sphere_d4"].Values, cam.State, 160, 96)
    assert_frames_identical(np.fromfile(raw, dtype=np.float32).reshape(96, 160, 4), ref, "C++ host Draw")
    d = np.fromfile(disp, dtype=np.uint8).reshape(96, 160, 4)
    dd = np.abs(d.astype(np.int16) - oracle_mod.display(ref).astype(np.int16))
    assert dd.max() <= 1 and (dd == 0).mean() >= 0.999
    # a mesh: MakeData imports the .ply, builds the ASDF on the GPU at Model.MaxDepth and caches
    # <mesh>.asdf next to it (Program.cs:638-641)
    write_ply(tmp_path / "ball.ply", fib_sphere(3000))
    out = subprocess.run([exe, str(tmp_path / "ball.ply"), "64", "64", str(raw)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cached = sb.OctData.LoadAsdf(str(tmp_path / "ball.asdf"))
    o = oracle_mod.sdfgen(fib_sphere(3000), 10)
    assert (cached.Structs == o["structs"]).all() and (cached.Values == o["values"]).all()
    ref, _ = oracle_mod.render(cached.Structs, cached.Values, sb.Logic(64, 64).State, 64, 64, nthreads=8)
    assert_frames_identical(np.fromfile(raw, dtype=np.float32).reshape(64, 64, 4), ref, "C++ host Draw of the built mesh")
