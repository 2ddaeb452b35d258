# Aggregate MFMA throughput over time from a compact kernel trace (see fused_trace.py): every kernel's flops (from its
# workgroup count and shape) spread uniformly over its duration, summed per window.
import gzip, sys, collections
rows = []
for ln in gzip.open(sys.argv[1], 'rt'):
    p = ln.split(None, 4)
    rows.append(dict(s=int(p[0]) / 10.0, d=int(p[1]) / 10.0, q=p[2], b=int(p[3]), n=p[4].strip()))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
W = float(sys.argv[3]) if len(sys.argv) > 3 else 4000.0
ends = [i for i, r in enumerate(rows) if r['n'].startswith('col_reduce')]
t0 = rows[ends[k - 1] + 1]['s'] if k > 0 else rows[0]['s']
t1 = rows[ends[k]]['s'] + rows[ends[k]]['d'] + 0.05
def flops(r):
    n, b = r['n'], r['b']
    if n.startswith('gemm_f64_kernel') or n.startswith('gemm3_f64_kernel'):
        K = 128 if n.startswith('gemm_f64_kernel<false, false, 2>') else 512
        return b * 2.0 * 128 * 128 * K, 'gemm'
    if n.startswith('gemm64_f64_kernel'):
        K = 128 if '2, ' in n[len('gemm64_f64_kernel<false, false, '):] [:3] else 512
        return b * 2.0 * 64 * 64 * K, 'gemm64'
    if n.startswith('tile_solve'):
        return b * 32 * 128 * 128 * 2.0 * 3 * 0.5, 'tile_solve'     # triangular factors: half the MACs
    if n.startswith('panel_solve_kernel<4'):
        return b * 16 * (512 * 512 * 3 + 512 * 512) * 1.0, 'panel_solve'   # ~16 columns x (3 refined tile solves + in-panel updates)
    if n.startswith('potrf_tile'):
        return 128 ** 3 / 3.0 * 3, 'potrf'
    return 0.0, 'other'
sel = [r for r in rows if t0 <= r['s'] < t1]
nb = int((t1 - t0) / W) + 1
acc = [collections.defaultdict(float) for _ in range(nb)]
tot = collections.defaultdict(float)
for r in sel:
    f, kind = flops(r)
    tot[kind] += f
    if f == 0 or r['d'] <= 0: continue
    s, e = r['s'] - t0, r['s'] - t0 + r['d']
    i = int(s / W)
    while i < nb and i * W < e:
        lo, hi = max(s, i * W), min(e, (i + 1) * W)
        if hi > lo: acc[i][kind] += f * (hi - lo) / r['d']
        i += 1
print("total flops by kind:", {k_: f"{v:.3e}" for k_, v in tot.items()}, f"sum {sum(tot.values()):.3e}")
for i, a in enumerate(acc):
    print(f"{i * W / 1e3:6.1f} ms: " + "  ".join(f"{k_} {a[k_] / W / 1e6:6.1f}" for k_ in ('gemm', 'gemm64', 'tile_solve', 'panel_solve')) + f"   total {sum(a.values()) / W / 1e6:6.1f} TFLOP/s")
