"""ASan + UBSan, then TSan, over the host half of the library (builders incl. the threaded
subtree build, .asdf I/O, validation, camera)."""
import os
import subprocess

from conftest import REPO


def test_host_code_under_asan_ubsan(tmp_path):
    src = os.path.join(REPO, "sdfbox_amd", "csrc")
    exe = str(tmp_path / "host_sanitize")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "host_sanitize.cpp")] +
                          [os.path.join(src, f) for f in ("errors.cpp", "asdf_io.cpp", "camera.cpp", "scene_gen.cpp")] +
                          ["-o", exe, "-lpthread"])
    out = subprocess.run([exe, str(tmp_path / "x.asdf")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host sanitizer run ok" in out.stdout


def test_threaded_builder_under_tsan(tmp_path):
    src = os.path.join(REPO, "sdfbox_amd", "csrc")
    exe = str(tmp_path / "host_tsan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-ffp-contract=off",
                           "-I", os.path.join(REPO, "include"), os.path.join(REPO, "tests", "host_sanitize.cpp")] +
                          [os.path.join(src, f) for f in ("errors.cpp", "asdf_io.cpp", "camera.cpp", "scene_gen.cpp")] +
                          ["-o", exe, "-lpthread"])
    out = subprocess.run([exe, str(tmp_path / "y.asdf")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr
