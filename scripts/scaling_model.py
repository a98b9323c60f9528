"""A prediction to be falsified: ms per frame at 2 / 4 / 8 MI355X from constants MEASURED ON ONE (profiles/bench_lines/r04_*.json and
scripts/multi_fixed_cost.py), for the first run on a real multi-GPU node to check.  No GPU needed.

grouped mode (throughput; groups of G frames per launch and gather, four groups in flight):
    T(N) = (s * t1 + E) / N + c_group / (G * 4)
    t1       one GPU, frames in flight, ms per frame                          (bench line)
    s        a rank's march writing a sparse share instead of the frame       ((sharded one-rank line - E) / t1)
    E        rank 0's expansion of all shares into the frame                  (133 MB at 5.8 TB/s per 4K frame in groups of four, by pixels)
    the rank-0 weight balances its share against E, so E divides by N as well; c_group = 60 us of host work per group and rank
    (launch, copy, events), G = 4 (8 at N = 8), hidden behind the other groups unless a share takes less.  Links: a rank's share
    is 1.1-1.5 bytes per pixel / N over its own xGMI link (~100 GB/s assumed of 153 peak): 14 us per 4K frame at N = 8 -- beside the march.
one frame at a time (latency; every device works on this frame):
    T(N) = f(N) + max(chain, w1 / N) + E + link
    f(N)     the pipeline's fixed cost beyond a bare launch (launch, copy, cross-stream wait, expansion launch, wait): measured on a
             64x64 frame over device lists that name the one GPU N times (0.011 / 0.028 / 0.042 / 0.048 ms at N = 1 / 2 / 4 / 8) + the bare
             launch-and-wait of 0.048 ms
    chain    a frame cannot end before its longest wave does: ~100 dependent march steps of ~1 us = 0.10 ms, whatever N is
    w1       one frame alone on one GPU minus that chain's share: the work that does divide
path-traced mode (cfg-5): dense RGBA32F bands, 133 MB / N per link + 0.05 ms of band copies on rank 0."""
import json, os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line(name):
    return json.loads([l for l in open(os.path.join(REPO, "profiles", "bench_lines", f"r04_{name}.json")) if l.startswith("{")][-1])


t1 = {"1080p": line("1080p_default")["ms_per_step"], "4K": line("4k_default")["ms_per_step"], "cfg-5 (4K, 16 spp)": line("cfg5_4k_spp16")["ms_per_step"]}
lat1 = {"1080p": line("1080p_default")["latency_ms"], "4K": line("4k_default")["latency_ms"], "cfg-5 (4K, 16 spp)": line("cfg5_4k_spp16")["latency_ms"]}
px = {"1080p": 1920 * 1080, "4K": 3840 * 2160}
E4k = 0.0228
s = (line("sharded_1rank_nccl_4k")["ms_per_step"] - E4k) / t1["4K"]
f = {1: 0.048 + 0.011, 2: 0.048 + 0.028, 4: 0.048 + 0.042, 8: 0.048 + 0.048}
chain, link_gbs = 0.10, 100.0
print(f"constants: s = {s:.3f}, E(4K) = {E4k} ms, chain = {chain} ms, link = {link_gbs} GB/s assumed, fixed cost f(N) = {f}")
print("| workload | mode | 1 GPU (measured) | 2 GPUs | 4 GPUs | 8 GPUs | speed-up at 8 |")
print("|---|---|---|---|---|---|---|")
for w in ("1080p", "4K"):
    E = E4k * px[w] / px["4K"]
    row = [f"{(s * t1[w] + E) / n + 0.060 / ((8 if n == 8 else 4) * 4):.4f}" for n in (2, 4, 8)]
    print(f"| {w} | groups of frames in flight | {t1[w]:.4f} | " + " | ".join(row) + f" | {t1[w] / float(row[-1]):.1f}x |")
    share_mb = 1.3 * px[w] / 1e6
    row = [f"{f[n] + max(chain, (lat1[w] - chain) / n) + E + share_mb / n / link_gbs:.3f}" for n in (2, 4, 8)]
    print(f"| {w} | one frame at a time | {lat1[w]:.3f} | " + " | ".join(row) + f" | {lat1[w] / float(row[-1]):.1f}x |")
w = "cfg-5 (4K, 16 spp)"
row = [f"{t1[w] / n + 133.0 / n / link_gbs + 0.05:.2f}" for n in (2, 4, 8)]
print(f"| {w} | frames in flight | {t1[w]:.2f} | " + " | ".join(row) + f" | {t1[w] / float(row[-1]):.1f}x |")
row = [f"{lat1[w] / n + 133.0 / n / link_gbs + 0.05 + f[n]:.2f}" for n in (2, 4, 8)]
print(f"| {w} | one frame at a time | {lat1[w]:.2f} | " + " | ".join(row) + f" | {lat1[w] / float(row[-1]):.1f}x |")
