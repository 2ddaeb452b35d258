"""A model of `bench.py --gpus P` (weak scaling) from rates measured on ONE MI355X and ASSUMED link figures -- a prediction to
hold against the driver's SCALE_rNN.json, nothing more.  Per panel of 512 columns the panel stream runs
    diagonal block (owner) -> broadcast -> rows below (Pr-fold parallel) -> gather -> look-ahead update (a)
while the update stream applies the previous panel to the rest (b); a step of the factorisation costs max(chain, b).
The solves stream the factor: per panel a gather and a local update of the rank's own right-hand-side columns.
Measured inputs (profiles/r02_*): tile Cholesky 38 us, small tile solve 14 us, small update 9 us, fused
panel chain 95 us per wave of workgroups, trailing update 50 TFLOP/s in situ (K = 512), small updates 40 TFLOP/s, copy 2 TB/s.
Assumed: LINK_GBS per direction per peer link, LAT_US per exchange."""
import sys

LINK_GBS = float(sys.argv[1]) if len(sys.argv) > 1 else 50.0
LAT_US = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
RATE, RATE_SMALL, COPY = 50e12, 40e12, 2e12
T_POTRF, T_SOLVE1, T_UPD1, T_FUSED = 38e-6, 14e-6, 9e-6, 95e-6
TILE, NBT = 128, 4


def xfer(bytes_per_link):
    return LAT_US * 1e-6 + bytes_per_link / (LINK_GBS * 1e9)


def step_time(n_side, P, Pr, Pc, split=False):
    N = n_side * n_side + 4 * n_side
    M = (n_side // 2) ** 2
    T = -(-N // TILE)
    comm = P > 1
    # ---- factorisation ----
    chain, b, tail = [], [], []
    for p0 in range(0, T, NBT):
        p1 = min(T, p0 + NBT)
        w = p1 - p0
        r = T - p1                                            # tile rows below the panel
        t_diag = w * T_POTRF + (w - 1) * (T_SOLVE1 + T_UPD1)
        t_bc = xfer((w * TILE) ** 2 * 8 + w * TILE * TILE * 8) if comm else 0.0
        rows_local = r * TILE / Pr
        t_rows = -(-int(rows_local / 16) // 512) * T_FUSED if r else 0.0
        S = r * TILE * w * TILE * 8.0
        t_gather = (xfer(S / Pr) + 2 * S / COPY) if comm else S / COPY
        t_tail = 0.0
        if split and comm and Pc == 1:
            # head: own rows copied locally + the next diagonal block's rows from their owner; the rest (tail) beside the chain
            t_tail = t_gather
            t_gather = xfer((w * TILE) ** 2 * 8.0) + 2 * (S / Pr) / COPY
        tail.append(t_tail)
        t_a = 2.0 * (r * TILE / Pr) * (w * TILE / Pc if Pc > 1 else w * TILE) * (w * TILE) / RATE_SMALL if r else 0.0
        chain.append(t_diag + t_bc + t_rows + t_gather + t_a)
        rr = max(r - NBT, 0)
        b.append((rr * (rr + 1) / 2.0) * 2.0 * TILE * TILE * (w * TILE) / P / RATE)
    t_fact = chain[0] + sum(max(chain[i + 1], b[i], tail[i]) for i in range(len(chain) - 1)) + b[-1]
    # ---- prediction: forward substitution of M / P columns per rank, factor streamed ----
    cols = M / P
    t_pred = 0.0
    for p0 in range(0, T, NBT):
        p1 = min(T, p0 + NBT)
        w, r = p1 - p0, T - p1
        S = (r + w) * TILE * w * TILE * 8.0
        t_g = (xfer(S / Pr) + 2 * S / COPY) if comm else 0.0
        t_fused = -(-int(cols / 16) // 512) * T_FUSED
        t_upd = 2.0 * r * TILE * cols * w * TILE / RATE
        t_pred += max(t_g + t_fused, t_upd)
    flops = N**3 / 3.0 + 2.0 * N * N + N * N * M + 4.0 * N * M
    t = t_fact + t_pred + 1.0e-3                                  # + assembly, boundary conditionings, read-outs
    return N, M, t, flops


sides = {1: 128, 2: 144, 4: 162, 8: 182}
base = None
print(f"assumed link {LINK_GBS:.0f} GB/s per direction per peer, {LAT_US:.0f} us per exchange")
for P, grids in ((1, [(1, 1)]), (2, [(2, 1)]), (4, [(4, 1), (2, 2)]), (8, [(8, 1), (2, 4), (4, 2)])):
    for Pr, Pc in grids:
        for split in ((False, True) if (P > 1 and Pc == 1) else (False,)):
            N, M, t, fl = step_time(sides[P], P, Pr, Pc, split)
            v = fl / t / 1e12
            if P == 1:
                base = v
            print(f"P={P} grid {Pr}x{Pc}{' split gather' if split else '':13s}: N_tot={N} M={M}  {t * 1e3:7.1f} ms  {v:6.1f} TFLOP/s  "
                  f"efficiency vs P x (1 GPU) = {v / (P * base):.2f}")
