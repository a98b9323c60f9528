"""The independent numpy restatement of SdfBox/Shaders/Compute.hlsl (tests/py_restatement.py), vectorised
over pixels so that whole frames -- thousands of pixels per scene and camera, silhouettes, shadowed and
100-step pixels included -- can be cross-checked against the C oracle, not a handful.

Written from the HLSL text like its scalar twin: every lane of the arrays is one invocation of main(),
loops run while any lane is in them, and a lane's state changes only under its own mask.  The value
texture is the reference's swizzled R8 texture (Program.cs:514-538) behind a bilinear sampler.  Every
operation is one float32 operation; the contract's fused multiply-adds (lerp, dot, pos += dir*s) are
computed in float64 -- the product of two float32 is exact there -- and rounded ONCE to float32: the sum is
split into its rounded value and its exact error (two-sum), and where the float64 sum sits exactly on a
float32 rounding tie the error decides the direction, so the result is the correctly rounded fma."""
import numpy as np

f32 = np.float32
TEXW = 8192  # Compute.hlsl:6


def fma(a, b, c):
    a = np.asarray(a, f32); b = np.asarray(b, f32); c = np.asarray(c, f32)
    with np.errstate(all="ignore"):
        p = a.astype(np.float64) * b.astype(np.float64)          # exact: 24 + 24 significant bits
        cd = c.astype(np.float64)
        s = p + cd                                                # one float64 rounding
        bb = s - p
        err = (p - (s - bb)) + (cd - bb)                          # its exact error (two-sum): the true value is s + err
        r = s.astype(f32)                                         # round to nearest even: RN32(s)
        # RN32(s + err) differs from RN32(s) only when s sits exactly on the midpoint of two adjacent float32
        # (|err| is at most half a float64 ulp of s: it cannot carry s across anything else): then the side of err wins
        back = r.astype(np.float64)
        other = np.where(s > back, np.nextafter(r, f32(np.inf)), np.nextafter(r, f32(-np.inf)))
        tie = np.isfinite(s) & np.isfinite(other) & (s != back) & (s == (back + other.astype(np.float64)) / 2) & (err != 0)
        hi, lo = np.maximum(r, other), np.minimum(r, other)
        return np.where(tie, np.where(err > 0, hi, lo), r).astype(f32)


def sat(x):
    with np.errstate(all="ignore"):
        return np.where(np.isnan(x), f32(0), np.minimum(np.maximum(x, f32(0)), f32(1))).astype(f32)


def lerp(a, b, t):
    return fma(t, (b - a).astype(f32), a)


def dot(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], (a[0] * b[0]).astype(f32)))


def normalize(v):
    with np.errstate(all="ignore"):
        r = (f32(1) / np.sqrt(dot(v, v))).astype(f32)
        return [(v[0] * r).astype(f32), (v[1] * r).astype(f32), (v[2] * r).astype(f32)]


class ShaderV:
    def __init__(self, structs, values, info):
        self.S = np.asarray(structs, np.int64).reshape(-1, 2)
        self.V = np.asarray(values, np.uint8).reshape(-1, 8)
        self.n = len(self.S)
        i = np.frombuffer(bytes(info), dtype=np.float32)
        self.heading = [[i[0], i[1], i[2]], [i[4], i[5], i[6]], [i[8], i[9], i[10]]]
        self.position = [i[12], i[13], i[14]]
        self.margin = i[15]
        self.screen = [i[16], i[17]]
        self.limit = i[19]
        self.light = [i[20], i[21], i[22]]
        self.strength = i[23]
        self.fov = i[24]

    # ---- the R8_UNorm value texture behind a bilinear sampler (Program.cs:514-538) --------------------
    def texel(self, tx, ty):
        node = (ty // 2) * (TEXW // 4) + tx // 4
        corner = np.array([[0, 1, 4, 5], [2, 3, 6, 7]])[ty % 2, tx % 4]
        ok = node < self.n
        b = self.V[np.where(ok, node, 0), corner]
        return np.where(ok, (b.astype(f32) / f32(255)).astype(f32), f32(0)).astype(f32)

    def sample(self, bx, by, wx, wy):
        top = lerp(self.texel(bx, by), self.texel(bx + 1, by), wx)
        bot = lerp(self.texel(bx, by + 1), self.texel(bx + 1, by + 1), wx)
        return lerp(top, bot, wy)

    def tex_origin(self):
        return (self.index * 4) % TEXW, self.index * 4 // TEXW * 2

    # ---- Cube (Compute.hlsl:31-58), under a lane mask ---------------------------------------------
    def inside(self, pos):
        hi = [(l + self.scale).astype(f32) for l in self.lower]
        with np.errstate(all="ignore"):
            r = np.ones(len(self.scale), bool)
            for l, p, h in zip(self.lower, pos, hi):
                r &= (l <= p) & (p <= h)
        return r

    def find(self, pos, m):
        """find(pos) for the lanes in mask m."""
        with np.errstate(all="ignore"):
            cpar = self.S[self.index, 0]; cchi = self.S[self.index, 1]
            self.nodes += m
            up = m & ~self.inside(pos) & (cpar >= 0)
            while up.any():
                self.index = np.where(up, cpar, self.index)
                cpar = np.where(up, self.S[self.index, 0], cpar); cchi = np.where(up, self.S[self.index, 1], cchi)
                self.nodes += up
                ns = (self.scale * f32(2)).astype(f32)
                self.scale = np.where(up, ns, self.scale)
                self.lower = [np.where(up, (np.floor((l / ns).astype(f32)) * ns).astype(f32), l) for l in self.lower]
                up = up & ~self.inside(pos) & (cpar >= 0)
            it = np.zeros(len(m), np.int64)
            dn = m & (self.index < self.n) & (it < 12) & (cchi >= 0)
            while dn.any():
                d = [sat((((p - l).astype(f32) / self.scale).astype(f32) * f32(2)).astype(f32)).astype(np.int64)
                     for p, l in zip(pos, self.lower)]
                self.index = np.where(dn, cchi + d[0] + 2 * d[1] + 4 * d[2], self.index)
                idx = np.where(dn, self.index, 0)
                cpar = np.where(dn, self.S[idx, 0], cpar); cchi = np.where(dn, self.S[idx, 1], cchi)
                self.nodes += dn
                ns = (self.scale / f32(2)).astype(f32)
                self.lower = [np.where(dn, (l + (di.astype(f32) * ns).astype(f32)).astype(f32), l) for l, di in zip(self.lower, d)]
                self.scale = np.where(dn, ns, self.scale)
                it += dn
                dn = dn & (self.index < self.n) & (it < 12) & (cchi >= 0)

    def local(self, pos):
        with np.errstate(all="ignore"):
            return [sat(((p - l).astype(f32) / self.scale).astype(f32)) for p, l in zip(pos, self.lower)]

    def interpol_world(self, pos, m):
        d = self.local(pos)
        self.samples += m
        ox, oy = self.tex_origin()
        loadL = self.sample(ox, oy, d[0], d[1])
        loadH = self.sample(ox + 2, oy, d[0], d[1])
        with np.errstate(all="ignore"):
            return (((lerp(loadL, loadH, d[2]) - f32(0.25)).astype(f32) * self.scale).astype(f32) * f32(2)).astype(f32)

    def gradient(self, pos):
        d = self.local(pos)
        ox, oy = self.tex_origin()
        z = np.zeros(len(self.scale), f32)
        sam = self.sample
        xl = lerp(sam(ox, oy, z, d[1]), sam(ox + 2, oy, z, d[1]), d[2])
        xh = lerp(sam(ox + 1, oy, z, d[1]), sam(ox + 3, oy, z, d[1]), d[2])
        yl = lerp(sam(ox, oy, d[0], z), sam(ox + 2, oy, d[0], z), d[2])
        yh = lerp(sam(ox, oy + 1, d[0], z), sam(ox + 2, oy + 1, d[0], z), d[2])
        zl = sam(ox, oy, d[0], d[1])
        zh = sam(ox + 2, oy, d[0], d[1])
        with np.errstate(all="ignore"):
            return [(xh - xl).astype(f32), (yh - yl).astype(f32), (zh - zl).astype(f32)]

    def ray(self, cx, cy):
        sx = ((cx.astype(f32) / self.screen[1]).astype(f32) - f32(f32(self.screen[0] / self.screen[1]) * f32(0.5))).astype(f32)
        sy = ((cy.astype(f32) / self.screen[1]).astype(f32) - f32(0.5)).astype(f32)
        v = [(sx * self.fov).astype(f32), (sy * self.fov).astype(f32), np.full(len(sx), 0.5, f32)]
        return normalize([dot(v, [np.full(len(sx), h, f32) for h in self.heading[j]]) for j in range(3)])

    def main(self, cx, cy):
        """main() for pixel arrays cx, cy -> (rgba [N, 4] float32, nodes [N], samples [N])."""
        with np.errstate(all="ignore"):
            return self._main(np.asarray(cx, np.int64), np.asarray(cy, np.int64))

    def _main(self, cx, cy):
        N = len(cx)
        self.index = np.zeros(N, np.int64)
        self.lower = [np.zeros(N, f32) for _ in range(3)]
        self.scale = np.ones(N, f32)
        self.nodes = np.zeros(N, np.int64); self.samples = np.zeros(N, np.int64)
        out = np.zeros((N, 4), f32)
        done = np.zeros(N, bool)
        pos = [np.full(N, p, f32) for p in self.position]
        d = self.ray(cx, cy)
        prox = np.ones(N, f32)
        m = self.margin
        i = np.zeros(N, np.int64)
        # for (i = 0; (prox > margin*2 || prox < 0) && i < 100; i++)
        run = ~done
        while True:
            run = run & ((prox > f32(m * f32(2))) | (prox < 0)) & (i < 100)
            if not run.any():
                break
            esc = run & (dot(pos, pos) > self.limit)
            out[esc] = np.stack([np.full(N, 0.005, f32), np.full(N, 0.01, f32), np.full(N, 0.2, f32), i.astype(f32)], 1)[esc]
            done |= esc
            run = run & ~esc
            if not run.any():
                break
            self.find(pos, run)
            pr = self.interpol_world(pos, run)
            prox = np.where(run, pr, prox)
            pos = [np.where(run, fma(di, prox, p), p) for p, di in zip(pos, d)]
            i += run
        live = ~done
        dl = normalize([(f32(l) - p).astype(f32) for l, p in zip(self.light, pos)])
        d = [np.where(live, a, b) for a, b in zip(dl, d)]
        pos = [np.where(live, fma(di, np.full(N, m, f32), p), p) for p, di in zip(pos, d)]
        angle = dot(d, normalize(self.gradient(pos)))
        back = live & (angle < 0)
        out[back] = np.stack([np.zeros(N, f32)] * 3 + [i.astype(f32)], 1)[back]
        done |= back
        diff = [(f32(l) - p).astype(f32) for l, p in zip(self.light, pos)]
        dist = (np.sqrt(dot(diff, diff)) / f32(2)).astype(f32)
        k = f32(f32(2.0 ** float(self.strength)) - f32(1))
        j = np.zeros(N, np.int64)
        self.shadow_rays = int((~done).sum())
        run = ~done
        while True:
            run = run & (j < 40) & (prox > -m)
            if not run.any():
                break
            lit = run & ((prox > dist) | (pos[0] < 0) | (pos[1] < 0) | (pos[2] < 0) | (pos[0] > 1) | (pos[1] > 1) | (pos[2] > 1))
            a = ((angle / (dist * dist).astype(f32)).astype(f32) * k).astype(f32)
            out[lit] = np.stack([a, a, a, (i + j).astype(f32)], 1)[lit]
            done |= lit
            run = run & ~lit
            near = run & (prox < m)
            if near.any():
                brk = near & (dot(self.gradient(pos), d) < 0)
                run = run & ~brk                       # break: falls through to the black write below
            if not run.any():
                break
            self.find(pos, run)
            pr = self.interpol_world(pos, run)
            prox = np.where(run, pr, prox)
            pos = [np.where(run, fma(di, (prox + m).astype(f32), p), p) for p, di in zip(pos, d)]
            j += run
        rest = ~done
        out[rest] = np.stack([np.zeros(N, f32)] * 3 + [(i + j).astype(f32)], 1)[rest]
        return out, self.nodes.copy(), self.samples.copy()
