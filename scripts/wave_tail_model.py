"""Dev script (GPU for the frame, numpy for the model): how many wave-iterations of the bench frame run with
few lanes alive, and what merging the tails of the four 8x8 waves of a 16x16 block would save."""
import sys
sys.path.insert(0, ".")
import numpy as np
import sdfbox_amd as sb
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080").split("x"))
od = sb.dragon_standin(9); sc = sb.Scene(od)
cam = sb.Logic(W, H); cam.Position = (0.5, 0.5, -0.35); cam.Heading = (-0.2, 0.35)
steps = sc.Draw(cam, W, H)[..., 3].astype(np.int32)
Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
s = np.zeros((Hp, Wp), np.int32); s[:H, :W] = steps
t = s.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3).reshape(Hp // 8, Wp // 8, 64)
t = np.sort(t, axis=2)[..., ::-1]                       # per tile: steps, descending
iters = t[..., 0].astype(np.int64)                      # wave iterations = its longest lane
total = iters.sum()
print(f"{W}x{H}: {t.shape[0] * t.shape[1]} waves, {total} wave-iterations, lane utilisation {s.sum() / (64 * total):.3f}")
for thresh in (32, 16, 8, 4):
    # iterations with <= thresh lanes alive = longest lane - the (thresh+1)-th longest
    tail = (t[..., 0] - t[..., thresh]).astype(np.int64)
    blocks = tail.reshape(Hp // 16, 2, Wp // 16, 2).transpose(0, 2, 1, 3).reshape(-1, 4)
    saved = blocks.sum(1) - blocks.max(1)
    print(f"  tails with <= {thresh:2d} lanes alive: {tail.sum() / total * 100:5.1f} % of the wave-iterations; "
          f"merging the four tails of a 16x16 block saves {saved.sum() / total * 100:5.1f} %")
