"""Camera state -> the `Info` block, mirroring SdfBox's `Logic` class.

`Logic.State`, `Logic.Heading`, `Logic.Position` (Logic.cs:30-78) keep their
names and meaning; the arithmetic lives in the native library (camera.cpp) so
that a C# / C caller of the same ABI gets the same bytes.
"""
import ctypes

from ._lib import Info, lib


class Logic:
    """One camera.  The reference keeps this state in statics; here it is an
    instance so that several frames/cameras can coexist."""

    xSize = 720   # Logic.cs:17-18
    ySize = 720

    def __init__(self, width=None, height=None):
        w = float(self.xSize if width is None else width)
        h = float(self.ySize if height is None else height)
        self.State = Info()
        lib.sdfhip_info_default(ctypes.byref(self.State), w, h)
        self._heading = (0.0, 0.0)
        self.mSpeed = 0.5        # Logic.cs:28

    # Logic.Heading, Logic.cs:46-55: (X = pitch, Y = yaw)
    @property
    def Heading(self):
        return self._heading

    @Heading.setter
    def Heading(self, value):
        hx, hy = float(value[0]), float(value[1])
        self._heading = (hx, hy)
        lib.sdfhip_info_set_heading(ctypes.byref(self.State), hx, hy)

    # Logic.Position, Logic.cs:60-78 (also refreshes State.limit)
    @property
    def Position(self):
        return tuple(self.State.position)

    @Position.setter
    def Position(self, value):
        lib.sdfhip_info_set_position(ctypes.byref(self.State), float(value[0]), float(value[1]),
                                     float(value[2]))

    # the camera part of Logic.Update, Logic.cs:239-272
    KEY_RIGHT, KEY_LEFT, KEY_UP, KEY_DOWN = 1, 2, 4, 8
    KEY_FORWARD, KEY_BACK, KEY_STRAFE_RIGHT, KEY_STRAFE_LEFT, KEY_SHIFT, KEY_CONTROL = 16, 32, 64, 128, 256, 512

    def Update(self, seconds, keys):
        """One time step with the pressed movement keys (a KEY_* bit mask)."""
        h = (ctypes.c_float * 2)(*self._heading)
        lib.sdfhip_camera_update(ctypes.byref(self.State), h, float(self.mSpeed), int(keys), float(seconds))
        self._heading = (h[0], h[1])

    def MouseMove(self, dx, dy):
        """Logic.MouseMove, Logic.cs:290-293."""
        h = (ctypes.c_float * 2)(*self._heading)
        lib.sdfhip_camera_mouse_move(ctypes.byref(self.State), h, float(dx), float(dy))
        self._heading = (h[0], h[1])

    def MouseWheel(self, delta):
        """The MouseWheel handler, Logic.cs:202-205."""
        self.mSpeed = lib.sdfhip_camera_mouse_wheel(float(self.mSpeed), float(delta))

    def Resize(self, width, height):
        """Program.Resize, Program.cs:282-292: screen_size follows the window."""
        self.State.screen_size[0] = float(width)
        self.State.screen_size[1] = float(height)

    def info_bytes(self):
        return bytes(self.State)
