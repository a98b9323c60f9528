"""First GPU contact: every kernel variant against the oracle (dev script)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdfbox_amd as s
import oracle

def cmp(name, gpu, ref):
    same = (gpu.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(gpu) & np.isnan(ref))
    bad = (~same).any(axis=-1).sum()
    print(f"  {name}: mismatching pixels {bad} / {gpu.shape[0]*gpu.shape[1]}", flush=True)
    if bad:
        ys, xs = np.nonzero((~same).any(axis=-1))
        for y, x in list(zip(ys, xs))[:5]:
            print("    ", x, y, gpu[y, x], ref[y, x])
    return bad

print("unorm table equal:", (s.unorm_table() == oracle.unorm_table()).all(), flush=True)
total_bad = 0
for scene_name, od in [("sphere_d4", s.sphere_d4()), ("torus_d6", s.torus_d6())]:
    sc = s.Scene(od)
    print(scene_name, od.Length, "depth", sc.depth, "stack ok", sc.stack_kernel_ok, flush=True)
    for (W, H, pos, head) in [(256, 256, (0.5, 0.5, 0.1), (0, 0)), (200, 120, (0.2, 0.3, -0.3), (-0.2, 0.35)),
                              (64, 64, (0.5, 0.5, 0.45), (0.3, 2.0))]:
        L = s.Logic(W, H); L.Position = pos; L.Heading = head
        ref, cnt = oracle.render(od.Structs, od.Values, L.State, W, H, nthreads=8)
        print(f" cam {pos} {head} {W}x{H} oracle counters {cnt}", flush=True)
        for kname, flags in [("generic", s.KERNEL_GENERIC), ("stack", s.KERNEL_STACK),
                             ("generic+compact", s.KERNEL_GENERIC | s.FLAG_COMPACT),
                             ("stack+compact", s.KERNEL_STACK | s.FLAG_COMPACT)]:
            img, st = s.Scene.Draw(sc, L, W, H, flags | s.FLAG_COUNT, want_stats=True)
            total_bad += cmp(kname, img, ref)
            ok = (st.n_nodes, st.n_samples, st.n_steps) == tuple(int(c) for c in cnt)
            print(f"    counters {st.n_nodes} {st.n_samples} {st.n_steps} match={ok} kernel_ms={st.kernel_ms:.3f}", flush=True)
            img2 = sc.Draw(L, W, H, flags)
            total_bad += cmp(kname + " (no count)", img2, ref)
    sc.close()
print("TOTAL BAD", total_bad)
sys.exit(1 if total_bad else 0)
