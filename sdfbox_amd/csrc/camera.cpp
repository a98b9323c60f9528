// Host-side construction of the `Info` block, mirroring SdfBox's Logic class.
//
//   sdfhip_info_default       Logic.State initialiser, Logic.cs:30-38, then
//                             Program.cs:54-55 (Heading = Heading; Position =
//                             Position) which fills heading and limit.
//   sdfhip_info_set_heading   Logic.Heading setter, Logic.cs:46-55, through
//                             Float3x3(Matrix4x4), Logic.cs:445-462.
//   sdfhip_info_set_position  Logic.Position setter, Logic.cs:60-78.
//   sdfhip_camera_update      the camera part of Logic.Update, Logic.cs:239-272 (arrow keys
//                             turn, WASD / numpad move in the yaw plane, shift / control
//                             move along y), with yawMat of Logic.cs:79-83.
//   sdfhip_camera_mouse_move  Logic.MouseMove, Logic.cs:290-293.
//   sdfhip_camera_mouse_wheel the MouseWheel handler, Logic.cs:202-205.
//
// Matrix4x4.CreateFromYawPitchRoll is .NET BCL code that is not in the
// reference tree; it is restated here from its published definition
// (Quaternion.CreateFromYawPitchRoll followed by Matrix4x4.CreateFromQuaternion).
// No reference test pins it; tests/test_camera.py pins the
// identity and quarter-turn cases by hand.
#include "abi_guard.h"
#include <cmath>
#include <cstring>

static void yaw_pitch_roll(float yaw, float pitch, float roll, float M[3][3])
{
    float sr = sinf(roll * 0.5f), cr = cosf(roll * 0.5f);
    float sp = sinf(pitch * 0.5f), cp = cosf(pitch * 0.5f);
    float sy = sinf(yaw * 0.5f), cy = cosf(yaw * 0.5f);
    float X = cy * sp * cr + sy * cp * sr;
    float Y = sy * cp * cr - cy * sp * sr;
    float Z = cy * cp * sr - sy * sp * cr;
    float W = cy * cp * cr + sy * sp * sr;
    float xx = X * X, yy = Y * Y, zz = Z * Z;
    float xy = X * Y, wz = Z * W, xz = Z * X, wy = Y * W, yz = Y * Z, wx = X * W;
    M[0][0] = 1.0f - 2.0f * (yy + zz); M[0][1] = 2.0f * (xy + wz);        M[0][2] = 2.0f * (xz - wy);
    M[1][0] = 2.0f * (xy - wz);        M[1][1] = 1.0f - 2.0f * (zz + xx); M[1][2] = 2.0f * (yz + wx);
    M[2][0] = 2.0f * (xz + wy);        M[2][1] = 2.0f * (yz - wx);        M[2][2] = 1.0f - 2.0f * (yy + xx);
}

extern "C" void sdfhip_info_set_heading(sdfhip_info *info, float heading_x, float heading_y)
try {
    if (!info) return;
    float M[3][3];
    // CreateFromYawPitchRoll(heading.Y, heading.X, 0), Logic.cs:53
    yaw_pitch_roll(heading_y, heading_x, 0.0f, M);
    // Float3x3(Matrix4x4 x): m11=M11 m12=M21 m13=M31 / m21=M12 ... Logic.cs:445-457
    for (int r = 0; r < 3; r++) {
        info->heading[r][0] = M[0][r];
        info->heading[r][1] = M[1][r];
        info->heading[r][2] = M[2][r];
        info->heading[r][3] = 0.0f;
    }
}
SDFHIP_ABI_CATCH_VOID(sdfhip_info_set_heading)

extern "C" void sdfhip_info_set_position(sdfhip_info *info, float x, float y, float z)
try {
    if (!info) return;
    info->position[0] = x; info->position[1] = y; info->position[2] = z;
    float furthest = 0.0f;
    for (int i = 0; i < 8; i++) {  // SdfMath.split(i), Math.cs:17-24
        float dx = (float)(i % 2) - x, dy = (float)(i / 2 % 2) - y, dz = (float)(i / 4 % 2) - z;
        float distance = dx * dx + dy * dy + dz * dz;  // Vector3.LengthSquared
        if (distance > furthest) furthest = distance;
    }
    info->limit = furthest;
}
SDFHIP_ABI_CATCH_VOID(sdfhip_info_set_position)

extern "C" void sdfhip_info_default(sdfhip_info *info, float width, float height)
try {
    if (!info) return;
    memset(info, 0, sizeof *info);
    info->light[0] = info->light[1] = info->light[2] = 0.0f;
    info->strength = 0.2f;
    info->margin = 0.0004f;
    info->screen_size[0] = width;
    info->screen_size[1] = height;
    info->fov = 1.0f;
    info->hidef = 0;
    sdfhip_info_set_heading(info, 0.0f, 0.0f);
    sdfhip_info_set_position(info, 0.5f, 0.5f, 0.1f);
}
SDFHIP_ABI_CATCH_VOID(sdfhip_info_default)

extern "C" void sdfhip_camera_update(sdfhip_info *info, float *heading_xy, float m_speed, uint32_t keys, float seconds)
try {
    if (!info || !heading_xy) return;
    const float t_speed = 0.1f;                      // tSpeed, Logic.cs:29
    const float transform = m_speed * m_speed * seconds;
    const float rotate = t_speed * seconds;
    float hx = heading_xy[0], hy = heading_xy[1];
    if (keys & SDFHIP_KEY_RIGHT) hy += rotate;       // Heading += (0, rotate)
    if (keys & SDFHIP_KEY_LEFT) hy -= rotate;
    if (keys & SDFHIP_KEY_UP) hx += rotate;          // Heading += (rotate, 0)
    if (keys & SDFHIP_KEY_DOWN) hx -= rotate;
    heading_xy[0] = hx; heading_xy[1] = hy;
    sdfhip_info_set_heading(info, hx, hy);
    float Y[3][3];
    yaw_pitch_roll(hy, 0.0f, 0.0f, Y);               // yawMat, Logic.cs:79-83
    float p[3] = { info->position[0], info->position[1], info->position[2] };
    // Position += Vector3.Transform(v, yawMat) * transform: row vector times matrix, then the scale, then the sum
    auto move = [&](float vx, float vy, float vz) {
        for (int j = 0; j < 3; j++) {
            float d = vx * Y[0][j] + vy * Y[1][j] + vz * Y[2][j];
            p[j] = p[j] + d * transform;
        }
    };
    if (keys & SDFHIP_KEY_FORWARD) move(0.0f, 0.0f, 1.0f);
    if (keys & SDFHIP_KEY_BACK) move(0.0f, 0.0f, -1.0f);
    if (keys & SDFHIP_KEY_STRAFE_RIGHT) move(1.0f, 0.0f, 0.0f);
    if (keys & SDFHIP_KEY_STRAFE_LEFT) move(-1.0f, 0.0f, 0.0f);
    if (keys & SDFHIP_KEY_SHIFT) p[1] = p[1] + -1.0f * transform;     // Position += (0, -1, 0) * transform
    if (keys & SDFHIP_KEY_CONTROL) p[1] = p[1] + 1.0f * transform;
    sdfhip_info_set_position(info, p[0], p[1], p[2]);
}
SDFHIP_ABI_CATCH_VOID(sdfhip_camera_update)

extern "C" void sdfhip_camera_mouse_move(sdfhip_info *info, float *heading_xy, float dx, float dy)
try {
    if (!info || !heading_xy) return;
    heading_xy[0] += -dy / 512.0f * 4.0f;             // Heading += new Vector2(-diff.Y, diff.X) / 512 * 4
    heading_xy[1] += dx / 512.0f * 4.0f;
    sdfhip_info_set_heading(info, heading_xy[0], heading_xy[1]);
}
SDFHIP_ABI_CATCH_VOID(sdfhip_camera_mouse_move)

extern "C" float sdfhip_camera_mouse_wheel(float m_speed, float wheel_delta)
try {
    if ((wheel_delta > 0 && m_speed < 1) || (wheel_delta < 0 && m_speed > 0.05))
        m_speed += wheel_delta * 0.05f;
    return m_speed;
}
SDFHIP_ABI_CATCH_AS(sdfhip_camera_mouse_wheel, 0.0f)
