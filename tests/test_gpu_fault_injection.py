"""The exception firewall of the DEVICE half of the library under injected allocation failures ("never throw/abort
across the boundary"; the host half: tests/test_host_sanitizers.py::test_injected_failures_come_back_as_status_codes).

The laboratory library's own operator new (lab.hip, sdfhip_debug_fail_host_allocations) throws std::bad_alloc at the k-th
allocation the library's host code makes: k = 0, 1, 2, ... until the call gets through without meeting the countdown.  Every
call must come back -- a status code with a message, or success -- the process must survive all of them, and the library must
still render the oracle's frame afterwards."""
import ctypes

import numpy as np
import pytest

from conftest import assert_frames_identical, make_camera

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lab():
    import sdfbox_amd.lab
    return sdfbox_amd.lab.load()


def _sweep(lab, what, call, limit=400):
    """call() under a countdown of 0, 1, 2, ...: -> (failures that came back as SdfHipError, allocations failed)"""
    hook = lab._lib.lib.sdfhip_debug_fail_host_allocations
    thrown = ctypes.c_uint64(0)
    hook(-1, ctypes.byref(thrown))
    errors = 0
    for k in range(limit):
        before = thrown.value
        hook(k, None)
        try:
            call()
            failed = None
        except lab.SdfHipError as e:
            failed = e
        finally:
            hook(-1, ctypes.byref(thrown))
        if thrown.value == before:                      # the countdown outlived the call: every allocation of it has been failed once
            assert failed is None, f"{what}: failed without an injected failure: {failed}"
            return errors, k
        if failed is not None:
            assert failed.code in (lab._lib.ERR_NOMEM, lab._lib.ERR_DEVICE), (what, k, failed)
            assert str(failed), (what, k)
            errors += 1
    raise AssertionError(f"{what}: still allocating after {limit} injected failures")


def test_every_gpu_entry_point_returns_a_code_when_an_allocation_fails(lab, oracle_mod, scenes):
    import torch
    od = scenes["sphere_d4"]
    W, H = 64, 48
    cam = make_camera("default", W, H)
    ref, _ = oracle_mod.render(od.Structs, od.Values, cam.State, W, H)
    report = {}
    made = []

    def upload():
        made.append(lab.Scene(od, device=0))
    report["sdfhip_scene_upload"] = _sweep(lab, "upload", upload)
    sc = made[-1]
    report["sdfhip_render"] = _sweep(lab, "render", lambda: sc.Draw(cam, W, H))
    buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    report["sdfhip_render_device (counting)"] = _sweep(
        lab, "render_device", lambda: (sc.DrawDevice(cam, W, H, buf.data_ptr(), flags=lab.FLAG_COUNT, stream=s, stats=lab.Stats()), torch.cuda.synchronize()))
    report["sdfhip_render_path"] = _sweep(lab, "render_path", lambda: sc.DrawPath(cam, W, H, pt=lab.PathTrace(spp=2)))
    report["sdfhip_render_display"] = _sweep(lab, "render_display", lambda: sc.DrawDisplay(cam, W, H))
    # the multi-device handle: worker threads, slots, a scene per rank, the link self-test -- two ranks on the one GPU
    multis = []

    def multi():
        m = lab.MultiScene(od, [0, 0])
        multis.append(m)
    report["sdfhip_multi_create"] = _sweep(lab, "multi_create", multi)
    ms = multis[-1]
    report["sdfhip_multi_render"] = _sweep(lab, "multi_render", lambda: ms.Draw(cam, W, H))
    # the point-cloud builder (its arenas, its level list) and the scene made from its tree
    i = np.arange(3000) + 0.5
    phi, th = np.arccos(1 - 2 * i / 3000), np.pi * (1 + 5 ** 0.5) * i
    p = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    cloud = np.concatenate([p * 0.5, p], 1).astype(np.float32)
    report["sdfhip_sdfgen"] = _sweep(lab, "sdfgen", lambda: lab.OctData.SdfGen(cloud, 4))
    fp = []
    report["sdfhip_sdfgen_scene"] = _sweep(lab, "sdfgen_scene", lambda: fp.append(lab.Scene.FromPoints(cloud, 4)))
    print({k: f"{v[1]} allocations failed one by one, {v[0]} came back as codes" for k, v in report.items()})
    # allocation failures did land in the device half, and each became a code
    # (upload and the render calls allocate with nothrow new / malloc / hipMalloc and check the result: nothing of theirs can throw;
    # the multi-device handle's vectors and threads and the builder's arenas and level list can)
    assert report["sdfhip_multi_create"][1] >= 3 and report["sdfhip_multi_render"][1] >= 1 and report["sdfhip_sdfgen"][1] >= 1, report
    assert sum(v[0] for v in report.values()) >= 10, report
    # ... and the library is whole: the same frames as before, from the handles that survived and from a fresh one
    assert_frames_identical(sc.Draw(cam, W, H), ref, "the scene that was uploaded under injection")
    assert_frames_identical(ms.Draw(cam, W, H), ref, "the multi-device handle that was created under injection")
    with lab.Scene(od, device=0) as fresh:
        assert_frames_identical(fresh.Draw(cam, W, H), ref, "a fresh scene after the sweeps")
    for h in made + multis + fp:
        h.close()


def test_the_product_has_no_injector(sb):
    assert not hasattr(sb._lib.lib, "sdfhip_debug_fail_host_allocations")
