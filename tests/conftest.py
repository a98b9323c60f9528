import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits_equal(a, b):
    """Bitwise equality of two float32 arrays, NaN == NaN regardless of payload."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def assert_frames_identical(got, ref, what=""):
    same = bits_equal(got, ref)
    if not same.all():
        bad = (~same).any(axis=-1)
        ys, xs = np.nonzero(bad)
        first = [(int(x), int(y), got[y, x].tolist(), ref[y, x].tolist()) for y, x in list(zip(ys, xs))[:4]]
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} pixels differ bitwise; first: {first}")


# The cameras of the golden frames (SURVEY.md 8c fixture list): default, rotated
# outside view, close-up inside the cube.  (position, (heading_x=pitch, heading_y=yaw))
CAMERAS = {
    "default": ((0.5, 0.5, 0.1), (0.0, 0.0)),
    "rotated": ((0.2, 0.3, -0.3), (np.deg2rad(-20.0), np.deg2rad(30.0))),
    "closeup": ((0.5, 0.42, 0.3), (0.3, 2.0)),
}


def make_camera(name, width, height):
    import sdfbox_amd as sb
    pos, head = CAMERAS[name]
    cam = sb.Logic(width, height)
    cam.Position = pos
    cam.Heading = head
    return cam


@pytest.fixture(scope="session")
def sb():
    import sdfbox_amd
    return sdfbox_amd


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def scenes():
    # (host arrays: the same for both flavours of the library -- test files that run on both override `sb` per module)
    import sdfbox_amd
    return {"sphere_d4": sdfbox_amd.sphere_d4(), "torus_d6": sdfbox_amd.torus_d6()}
