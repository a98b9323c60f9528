"""`Info` construction against Logic.cs:30-78,427-463.  Matrix4x4.CreateFromYawPitchRoll
is .NET BCL code outside the reference tree (unpinned): identity
and quarter turns are checked by hand, the general case against an independent
float64 composition of axis rotations."""
import numpy as np


def heading_rows(info):
    return np.array([[info.heading[r][c] for c in range(3)] for r in range(3)], dtype=np.float64)


def transform(v, info):
    # mul(v, inf.heading) as the shader sees it: component j = dot(v, row j)  
    return heading_rows(info) @ np.asarray(v, dtype=np.float64)


def test_defaults(sb):
    cam = sb.Logic()
    s = cam.State
    assert tuple(s.position) == (0.5, 0.5, np.float32(0.1))
    assert tuple(s.light) == (0.0, 0.0, 0.0)
    assert s.strength == np.float32(0.2) and s.margin == np.float32(0.0004)
    assert tuple(s.screen_size) == (720.0, 720.0) and s.fov == 1.0 and s.hidef == 0
    assert np.allclose(heading_rows(s), np.eye(3))
    assert [s.heading[r][3] for r in range(3)] == [0.0, 0.0, 0.0]
    # limit = max squared distance to a unit-cube corner = 0.25 + 0.25 + 0.81
    assert abs(s.limit - 1.31) < 1e-6
    assert len(cam.info_bytes()) == 112


def test_position_updates_limit(sb):
    cam = sb.Logic(64, 64)
    cam.Position = (0.5, 0.5, -0.35)
    assert abs(cam.State.limit - (0.25 + 0.25 + 1.35 ** 2)) < 1e-6
    cam.Position = (2.0, -1.0, 0.25)
    assert abs(cam.State.limit - (4.0 + 4.0 + 0.75 ** 2)) < 1e-6


def test_quarter_turns(sb):
    cam = sb.Logic(64, 64)
    # Heading = (X = pitch, Y = yaw): Logic.cs:53.  Vector3.Transform((0,0,1), yaw 90 deg) = (1,0,0)
    cam.Heading = (0.0, np.pi / 2)
    assert np.allclose(transform([0, 0, 1], cam.State), [1, 0, 0], atol=1e-6)
    assert np.allclose(transform([1, 0, 0], cam.State), [0, 0, -1], atol=1e-6)
    assert np.allclose(transform([0, 1, 0], cam.State), [0, 1, 0], atol=1e-6)
    # pitch 90 deg: rotation about x; (0,0,1) -> (0,-1,0), (0,1,0) -> (0,0,1)
    cam.Heading = (np.pi / 2, 0.0)
    assert np.allclose(transform([0, 0, 1], cam.State), [0, -1, 0], atol=1e-6)
    assert np.allclose(transform([0, 1, 0], cam.State), [0, 0, 1], atol=1e-6)


def test_general_heading_against_axis_rotations(sb):
    # .NET: yaw about y, pitch about x, roll about z; row-vector convention
    # v' = v * (Rz(roll) * Rx(pitch) * Ry(yaw)); roll = 0 here.
    def rx(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[1, 0, 0], [0, c, s], [0, -s, c]])

    def ry(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])

    cam = sb.Logic(64, 64)
    rng = np.random.default_rng(7)
    for _ in range(20):
        pitch, yaw = rng.uniform(-3, 3, 2)
        cam.Heading = (pitch, yaw)
        M = rx(pitch) @ ry(yaw)                       # Matrix4x4 M, v' = v M
        for v in rng.normal(size=(3, 3)):
            assert np.allclose(transform(v, cam.State), v @ M, atol=2e-6)
        # Float3x3 stores columns of M as rows (Logic.cs:445-457)
        assert np.allclose(heading_rows(cam.State), M.T, atol=2e-6)


def test_info_blocks_match_golden(sb):
    import os
    from conftest import CAMERAS, GOLDEN, make_camera
    g = np.load(os.path.join(GOLDEN, "info_blocks.npz"))
    for name in CAMERAS:
        assert bytes(make_camera(name, 64, 64).State) == g[name].tobytes(), name


def test_update_moves_as_logic_update_does(sb):
    # Logic.Update, Logic.cs:239-272: transform = mSpeed^2 * sec, rotate = tSpeed * sec (0.1)
    cam = sb.Logic(64, 64)
    assert cam.mSpeed == 0.5
    p0 = np.array(cam.Position, dtype=np.float64)
    cam.Update(2.0, cam.KEY_FORWARD)                      # W: +z by 0.25 * 2
    assert np.allclose(cam.Position, p0 + [0, 0, 0.5], atol=1e-7)
    cam.Update(2.0, cam.KEY_BACK | cam.KEY_STRAFE_RIGHT)  # S and D: back where it was in z, +x by 0.5
    assert np.allclose(cam.Position, p0 + [0.5, 0, 0], atol=1e-7)
    cam.Update(1.0, cam.KEY_SHIFT)                        # LShift: Position += (0, -1, 0) * 0.25
    assert np.allclose(cam.Position, p0 + [0.5, -0.25, 0], atol=1e-7)
    cam.Update(1.0, cam.KEY_CONTROL | cam.KEY_STRAFE_LEFT)
    assert np.allclose(cam.Position, p0 + [0.25, 0, 0], atol=1e-7)
    # the Position setter ran: limit follows
    x, y, z = cam.Position
    assert abs(cam.State.limit - max((cx - x) ** 2 + (cy - y) ** 2 + (cz - z) ** 2
                                     for cx in (0, 1) for cy in (0, 1) for cz in (0, 1))) < 1e-6
    # arrow keys turn: Right = yaw += 0.1 * sec, Up = pitch += 0.1 * sec; movement uses the NEW yaw, never the pitch
    cam = sb.Logic(64, 64)
    cam.Update(float(np.pi / 2 / 0.1), cam.KEY_RIGHT)     # yaw = 90 degrees
    assert np.allclose(cam.Heading, (0.0, np.pi / 2), atol=1e-6)
    assert np.allclose(transform([0, 0, 1], cam.State), [1, 0, 0], atol=1e-6)
    q0 = np.array(cam.Position, dtype=np.float64)
    cam.Update(4.0, cam.KEY_FORWARD | cam.KEY_UP)         # forward is now +x; pitching does not tilt the walk
    assert np.allclose(cam.Position, q0 + [1.0, 0, 0], atol=1e-6)
    assert np.allclose(cam.Heading, (0.4, np.pi / 2), atol=1e-6)
    cam.Update(1.0, cam.KEY_LEFT | cam.KEY_DOWN)
    assert np.allclose(cam.Heading, (0.3, np.pi / 2 - 0.1), atol=1e-6)


def test_mouse_move_and_wheel(sb):
    cam = sb.Logic(64, 64)
    cam.MouseMove(128.0, -64.0)                            # Heading += (-dy, dx) / 512 * 4
    assert np.allclose(cam.Heading, (0.5, 1.0), atol=1e-7)
    ref = sb.Logic(64, 64); ref.Heading = (0.5, 1.0)
    assert bytes(cam.State) == bytes(ref.State)
    cam.MouseWheel(1.0); assert abs(cam.mSpeed - 0.55) < 1e-7
    for _ in range(20):
        cam.MouseWheel(1.0)
    assert 1.0 <= cam.mSpeed < 1.05                        # grows only while mSpeed < 1
    for _ in range(40):
        cam.MouseWheel(-1.0)
    assert 0.0 < cam.mSpeed <= 0.05 + 1e-6                 # shrinks only while mSpeed > 0.05
