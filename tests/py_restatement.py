"""Second, independent restatement of SdfBox/Shaders/Compute.hlsl in numpy
float32 scalars (every operation individually rounded; the mads of lerp, dot
and pos += dir*s fused, computed exactly with rationals), written from the HLSL
text, not from oracle/sdf_oracle.c.  Slow; used on a handful of pixels to
cross-check the C oracle bit for bit (tests/test_oracle.py)."""
from fractions import Fraction

import numpy as np

f32 = np.float32
TEXW = 8192  # Compute.hlsl:6


def sat(x):
    # HLSL saturate: NaN -> 0
    if np.isnan(x):
        return f32(0)
    return f32(min(max(x, f32(0)), f32(1)))


def fma(a, b, c):
    """Correctly rounded float32 a*b + c (one rounding), by exact rational arithmetic."""
    a, b, c = f32(a), f32(b), f32(c)
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        return f32(f32(a * b) + c)                      # inf/nan propagate the same way
    v = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    if v == 0:
        # sign of an exact zero sum: +0 unless both addends are -0
        prod_neg = (np.signbit(a) != np.signbit(b))
        return f32(-0.0) if (prod_neg and np.signbit(c)) else f32(0.0)
    sign = -1 if v < 0 else 1
    m = abs(v)
    e = m.numerator.bit_length() - m.denominator.bit_length()   # 2^e <= m < 2^(e+2) roughly
    while Fraction(2) ** e > m:
        e -= 1
    while Fraction(2) ** (e + 1) <= m:
        e += 1
    q = max(e - 23, -149)                                # ulp exponent (subnormals: 2^-149)
    scaled = m / (Fraction(2) ** q)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    return f32(sign * float(n) * 2.0 ** q)               # n < 2^25 and 2^q are exact in double


def lerp(a, b, t):
    # HLSL lerp compiles to add + mad; the contract fuses the mad (DESIGN.md "Numerics")
    return fma(t, f32(b - a), a)


def dot(a, b):
    # dp3 as a chain of mads
    return fma(a[2], b[2], fma(a[1], b[1], f32(a[0] * b[0])))


def normalize(v):
    # HLSL normalize(v) = v * rsqrt(dot(v, v)); rsqrt(x) = 1 / sqrt(x), each correctly rounded
    r = f32(f32(1) / np.sqrt(dot(v, v)))
    return [f32(v[0] * r), f32(v[1] * r), f32(v[2] * r)]


class Texture:
    """The R8_UNorm value texture of Program.cs:514-538 with a bilinear sampler:
    node i -> texels x in [4i % 8192, +4), rows 2*(4i // 8192) and +1; row0 =
    corners {0,1,4,5}, row1 = {2,3,6,7}."""

    def __init__(self, values):
        self.values = values

    def texel(self, tx, ty):
        # inverse of the swizzle: which node/corner sits at texel (tx, ty)
        node = (ty // 2) * (TEXW // 4) + tx // 4
        col, row = tx % 4, ty % 2
        corner = [[0, 1, 4, 5], [2, 3, 6, 7]][row][col]
        if node >= len(self.values):
            return f32(0)
        return f32(f32(self.values[node][corner]) / f32(255))

    def sample(self, bx, by, wx, wy):
        """One bilinear tap between texels (bx, by) .. (bx+1, by+1) with weights
        (wx, wy).  sam(p) of Compute.hlsl:15-18 samples at texel-space p + 0.5, i.e.
        base = floor(p), weight = frac(p); the weights are taken from d *before*
        the integer texel origin is added (DESIGN.md "Numerics": the shader adds
        the origin in fp32 first, which only coarsens the weights)."""
        top = lerp(self.texel(bx, by), self.texel(bx + 1, by), wx)
        bot = lerp(self.texel(bx, by + 1), self.texel(bx + 1, by + 1), wx)
        return lerp(top, bot, wy)


class Shader:
    def __init__(self, structs, values, info):
        self.S = structs
        self.tex = Texture(values)
        self.n = len(structs)
        i = np.frombuffer(bytes(info), dtype=np.float32)
        self.heading = [[i[0], i[1], i[2]], [i[4], i[5], i[6]], [i[8], i[9], i[10]]]
        self.position = [i[12], i[13], i[14]]
        self.margin = i[15]
        self.screen = [i[16], i[17]]
        self.limit = i[19]
        self.light = [i[20], i[21], i[22]]
        self.strength = i[23]
        self.fov = i[24]
        self.nodes = 0
        self.samples = 0

    # Cube
    def scale_up(self):
        self.scale = f32(self.scale * f32(2))
        self.lower = [f32(np.floor(f32(l / self.scale)) * self.scale) for l in self.lower]

    def scale_down(self, p):
        self.scale = f32(self.scale / f32(2))
        self.lower = [f32(l + f32(f32(pi) * self.scale)) for l, pi in zip(self.lower, p)]

    def inside(self, pos):
        hi = [f32(l + self.scale) for l in self.lower]
        return all(l <= p for l, p in zip(self.lower, pos)) and all(p <= h for p, h in zip(pos, hi))

    def tex_origin(self):
        return (self.index * 4) % TEXW, self.index * 4 // TEXW * 2

    def sample_at(self, d, scale):
        ox, oy = self.tex_origin()
        loadL = self.tex.sample(ox, oy, d[0], d[1])          # sam(d.xy + p)
        loadH = self.tex.sample(ox + 2, oy, d[0], d[1])      # sam(d.xy + p + float2(2, 0))
        return f32(f32(f32(lerp(loadL, loadH, d[2]) - f32(0.25)) * scale) * f32(2))

    def interpol_world(self, pos):
        d = [sat(f32(f32(p - l) / self.scale)) for p, l in zip(pos, self.lower)]
        self.samples += 1
        return self.sample_at(d, self.scale)

    def find(self, pos):
        iterations = 0
        c = self.S[self.index]; self.nodes += 1
        while (not self.inside(pos)) and c[0] >= 0:
            self.index = int(c[0]); c = self.S[self.index]; self.nodes += 1
            self.scale_up()
        while self.index < self.n and iterations < 12 and c[1] >= 0:
            d = [int(sat(f32(f32(f32(p - l) / self.scale) * f32(2)))) for p, l in zip(pos, self.lower)]
            self.index = int(c[1]) + d[0] + 2 * d[1] + 4 * d[2]
            c = self.S[self.index]; self.nodes += 1
            self.scale_down(d)
            iterations += 1

    def gradient(self, pos):
        d = [sat(f32(f32(p - l) / self.scale)) for p, l in zip(pos, self.lower)]
        ox, oy = self.tex_origin()
        sam = self.tex.sample
        z = f32(0)
        # float2(0, d.y) + p / + ph, float2(1, d.y) + p / + ph   (Compute.hlsl:120-121)
        xl = lerp(sam(ox, oy, z, d[1]), sam(ox + 2, oy, z, d[1]), d[2])
        xh = lerp(sam(ox + 1, oy, z, d[1]), sam(ox + 3, oy, z, d[1]), d[2])
        # float2(d.x, 0) + p / + ph, float2(d.x, 1) + p / + ph   (Compute.hlsl:123-124)
        yl = lerp(sam(ox, oy, d[0], z), sam(ox + 2, oy, d[0], z), d[2])
        yh = lerp(sam(ox, oy + 1, d[0], z), sam(ox + 2, oy + 1, d[0], z), d[2])
        zl = sam(ox, oy, d[0], d[1])
        zh = sam(ox + 2, oy, d[0], d[1])
        return [f32(xh - xl), f32(yh - yl), f32(zh - zl)]

    def ray(self, cx, cy):
        sx = f32(f32(f32(cx) / self.screen[1]) - f32(f32(self.screen[0] / self.screen[1]) * f32(0.5)))
        sy = f32(f32(f32(cy) / self.screen[1]) - f32(0.5))
        v = [f32(sx * self.fov), f32(sy * self.fov), f32(0.5)]
        return normalize([dot(v, self.heading[j]) for j in range(3)])

    def main(self, cx, cy):
        with np.errstate(all="ignore"):
            return self._main(cx, cy)

    def _main(self, cx, cy):
        self.index = 0
        self.lower = [f32(0), f32(0), f32(0)]
        self.scale = f32(1)
        self.nodes = self.samples = 0
        pos = [f32(p) for p in self.position]
        d = self.ray(cx, cy)
        prox = f32(1)
        m = self.margin
        i = 0
        while (prox > f32(m * f32(2)) or prox < 0) and i < 100:
            if dot(pos, pos) > self.limit:
                return [f32(0.005), f32(0.01), f32(0.2), f32(i)]
            self.find(pos)
            prox = self.interpol_world(pos)
            pos = [fma(di, prox, p) for p, di in zip(pos, d)]
            i += 1
        d = normalize([f32(l - p) for l, p in zip(self.light, pos)])
        pos = [fma(di, m, p) for p, di in zip(pos, d)]
        angle = dot(d, normalize(self.gradient(pos)))
        if angle < 0:
            return [f32(0), f32(0), f32(0), f32(i)]
        diff = [f32(l - p) for l, p in zip(self.light, pos)]
        dist = f32(np.sqrt(dot(diff, diff)) / f32(2))
        k = f32(f32(2.0 ** float(self.strength)) - f32(1))
        j = 0
        while j < 40 and prox > -m:
            if prox > dist or any(p < 0 for p in pos) or any(p > 1 for p in pos):
                a = f32(f32(angle / f32(dist * dist)) * k)
                return [a, a, a, f32(i + j)]
            if prox < m:
                if dot(self.gradient(pos), d) < 0:
                    break
            self.find(pos)
            prox = self.interpol_world(pos)
            pos = [fma(di, f32(prox + m), p) for p, di in zip(pos, d)]
            j += 1
        return [f32(0), f32(0), f32(0), f32(i + j)]
