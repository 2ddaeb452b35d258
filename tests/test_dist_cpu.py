"""Process-group plumbing (world_size 2 on CPU): the TCP control plane used by bench.py and
its equivalence with a torch.distributed gloo group on the same ranks."""
import multiprocessing as mp
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gloo_port, q):
    sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port - 1))
    from linpde_gp_amd import _dist
    comm = _dist.Comm.from_env()
    comm.barrier()
    mx = comm.allreduce_max(10.0 + rank)
    got = comm.bcast({"uid": b"\x01\x02"} if rank == 0 else None)
    ag = comm.allgather(rank * 2)
    # same collectives through torch.distributed / gloo
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{gloo_port}", rank=rank, world_size=world)
    t = torch.tensor([10.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    comm.close()
    q.put((rank, mx, got["uid"], ag, float(t[0])))


def test_comm_world2_matches_gloo():
    world = 2
    port, gloo_port = _free_port(), _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, gloo_port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, mx, uid, ag, gl in res:
        assert mx == 11.0 == gl and uid == b"\x01\x02" and ag == [0, 2]


def _plain_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
    from linpde_gp_amd import _dist
    comm = _dist.Comm(rank, world, "127.0.0.1", port)
    q.put((rank, comm.allgather(rank + 5)))
    comm.close()


def test_comm_survives_a_foreign_listener_on_its_port():
    """The control-plane port (MASTER_PORT + 1) may be taken: rank 0 moves to the next free port
    and the clients find it by handshake."""
    port = _free_port()
    squatter = socket.socket()
    squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    squatter.bind(("127.0.0.1", port))
    squatter.listen(4)                         # accepts connections, never answers the handshake
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_plain_worker, args=(r, 3, port, q)) for r in range(3)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res == [(0, [5, 6, 7]), (1, [5, 6, 7]), (2, [5, 6, 7])]
    finally:
        squatter.close()


def test_bench_weak_scaling_sizes():
    """bench.py grows the grid with the number of GPUs so that the algorithmic flops per GPU stay
    those of c3."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    def flops(n, m):
        N, M = float(n * n + 4 * n), float(m * m)
        return N**3 / 3 + 2 * N * N + N * N * M + 4 * N * M
    assert bench._weak_sides(1) == (128, 64)
    for w in (2, 4, 8):
        n, m = bench._weak_sides(w)
        assert m == n // 2 and abs(flops(n, m) / (w * flops(128, 64)) - 1.0) < 0.06


# ---- model of the distributed factorisation (csrc/dist.hip: potrf_dist) ---------------------------------
def _model_potrf_dist_2d(tiles, dblk, t_done, T, nbt, tb, pr, pc, rank, bcast):
    """NumPy mirror of potrf_dist, step for step, on a Pr x Pc grid with 2-D block-cyclic tiles: `tiles` maps the
    GLOBAL tile index (gi, gj), gi >= gj, to a tb x tb array and holds ONLY the tiles this rank owns; `dblk` is the
    replicated store of diagonal-block tiles.  Per panel: the owner of the diagonal block factors and broadcasts it,
    the panel's process column solves its rows, the rows below are gathered on every rank (one broadcast per source),
    every rank updates its own tiles.  A block append runs the old panels first (new rows only)."""
    import numpy as np
    import scipy.linalg as sla
    my_r, my_c = divmod(rank, pc)
    own_r = lambda g: (g // nbt) % pr
    own_c = lambda g: (g // nbt) % pc
    panels = []
    if t_done > 0 and T > t_done:
        c0 = 0
        while c0 < t_done:
            c1 = min(t_done, (c0 // nbt + 1) * nbt)
            panels.append((c0, c1, False))
            c0 = c1
    c0 = t_done
    while c0 < T:
        c1 = min(T, (c0 // nbt + 1) * nbt)
        panels.append((c0, c1, True))
        c0 = c1
    for c0, c1, fresh in panels:
        K, kw = c0 // nbt, c1 - c0
        owner = own_r(c0) * pc + own_c(c0)
        blk0 = K * nbt
        if fresh:
            if rank == owner:
                D = np.zeros((kw * tb, kw * tb))
                for i in range(kw):
                    for j in range(i + 1):
                        D[i * tb:(i + 1) * tb, j * tb:(j + 1) * tb] = tiles[(c0 + i, c0 + j)]
                Lb = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)
                for i in range(kw):
                    for j in range(i + 1):
                        tiles[(c0 + i, c0 + j)] = Lb[i * tb:(i + 1) * tb, j * tb:(j + 1) * tb].copy()
                # rows of this panel x the OLD columns of the same block (block append inside a block)
                send = np.stack([tiles[(gi, gj)] for gi in range(c0, c1) for gj in range(blk0, gi + 1)])
            else:
                send = np.empty((sum(gi + 1 - blk0 for gi in range(c0, c1)), tb, tb))
            got = bcast(send, owner)
            it = iter(got)
            for gi in range(c0, c1):
                for gj in range(blk0, gi + 1):
                    dblk[(gi, gj)] = next(it).copy()
        if c1 >= T:
            continue
        row_lo = c1 if fresh else t_done
        LKK = np.zeros((kw * tb, kw * tb))
        for i in range(kw):
            for j in range(i + 1):
                LKK[i * tb:(i + 1) * tb, j * tb:(j + 1) * tb] = dblk[(c0 + i, c0 + j)]
        LKK = np.tril(LKK)
        if my_c == own_c(c0):
            for gi in range(row_lo, T):
                if own_r(gi) != my_r:
                    continue
                A = np.hstack([tiles[(gi, gj)] for gj in range(c0, c1)])
                X = sla.solve_triangular(LKK, A.T, lower=True).T
                for j, gj in enumerate(range(c0, c1)):
                    tiles[(gi, gj)] = X[:, j * tb:(j + 1) * tb].copy()
        nxt = min(T, (c1 // nbt + 1) * nbt)          # end of the next panel (panels never straddle a block)
        if pc == 1 and pr > 1:
            # SPLIT GATHER (P x 1 grids, dist.hip): the look-ahead update of the next panel's columns [c1, nxt) may read this
            # rank's OWN rows and the rows of the next diagonal block ONLY -- `panel` holds nothing else at that point, so
            # any other access raises KeyError -- and the rest of the panel arrives afterwards
            panel = {gi: np.hstack([tiles[(gi, gj)] for gj in range(c0, c1)]) for gi in range(c1, T) if own_r(gi) == my_r}
            root = own_r(c1) * pc + own_c(c0)
            head = list(range(c1, nxt))
            if rank == root:
                send = np.stack([panel[gi] for gi in head])
            else:
                send = np.empty((len(head), tb, kw * tb))
            for gi, P in zip(head, bcast(send, root)):
                panel[gi] = P
            for gi in range(row_lo, T):              # (a)
                if own_r(gi) != my_r:
                    continue
                for gj in range(c1, min(nxt, gi + 1)):
                    tiles[(gi, gj)] = tiles[(gi, gj)] - panel[gi] @ panel[gj].T
            for r in range(pr):                      # tail: every member's rows beyond the next block
                rows = [gi for gi in range(nxt, T) if own_r(gi) == r]
                if not rows:
                    continue
                root = r * pc + own_c(c0)
                if rank == root:
                    send = np.stack([panel[gi] for gi in rows])
                else:
                    send = np.empty((len(rows), tb, kw * tb))
                for gi, P in zip(rows, bcast(send, root)):
                    panel.setdefault(gi, P)
            for gi in range(row_lo, T):              # (b)
                if own_r(gi) != my_r:
                    continue
                for gj in range(nxt, gi + 1):
                    tiles[(gi, gj)] = tiles[(gi, gj)] - panel[gi] @ panel[gj].T
            continue
        # gather the rows below the panel.  SCOPED (round 4, dist.hip: gather_rows): with Pr, Pc > 1 a rank receives only
        # the rows its updates read -- row blocks B with B % Pr == my_r (rows of its tiles) or B % Pc == my_c (columns of its
        # tiles); the pieces are the classes q = B mod lcm(Pr, Pc), each from rank (q % Pr, process column of the panel).
        # `panel` holds nothing else, so an update that read any other row raises KeyError.  (The transport here is a
        # gloo broadcast: a rank outside the destination set takes part in it and drops the piece.)
        import math
        panel = {}
        lc = math.lcm(pr, pc) if (pr > 1 and pc > 1) else pr
        for q in range(lc):
            rows = [gi for gi in range(c1, T) if (gi // nbt) % lc == q]
            if not rows:
                continue
            root = (q % pr) * pc + own_c(c0)
            if rank == root:
                send = np.stack([np.hstack([tiles[(gi, gj)] for gj in range(c0, c1)]) for gi in rows])
            else:
                send = np.empty((len(rows), tb, kw * tb))
            got = bcast(send, root)
            if lc == pr or my_r == q % pr or my_c == q % pc:
                for gi, P in zip(rows, got):
                    panel[gi] = P
        for gi in range(row_lo, T):
            if own_r(gi) != my_r:
                continue
            for gj in range(c1, gi + 1):
                if own_c(gj) == my_c:
                    tiles[(gi, gj)] = tiles[(gi, gj)] - panel[gi] @ panel[gj].T
    return tiles


def _dist_model_worker(rank, world, pr, pc, gloo_port, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{gloo_port}", rank=rank, world_size=world)

    def bcast(arr, root):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        dist.broadcast(t, src=root)
        return t.numpy()

    rng = np.random.default_rng(5)               # same matrix on every rank
    tb, nbt = 6, 2                               # scaled-down tiles; block = 2 tiles
    T1, T2, T3 = 5, 12, 15                       # three conditionings; 5 and 12 are NOT block-aligned, 15 is ragged
    n = T3 * tb
    M = rng.standard_normal((n, n + 10))
    G = M @ M.T + 0.5 * np.eye(n)
    my_r, my_c = divmod(rank, pc)
    mine = lambda gi, gj: (gi // nbt) % pr == my_r and (gj // nbt) % pc == my_c
    tiles, dblk = {}, {}
    ok = True
    for t_done, T in ((0, T1), (T1, T2), (T2, T3)):
        # "assemble": only the tiles this rank owns, only the new block row
        for gi in range(t_done, T):
            for gj in range(gi + 1):
                if mine(gi, gj):
                    tiles[(gi, gj)] = G[gi * tb:(gi + 1) * tb, gj * tb:(gj + 1) * tb].copy()
        tiles = _model_potrf_dist_2d(tiles, dblk, t_done, T, nbt, tb, pr, pc, rank, bcast)
        ref = np.linalg.cholesky(G[:T * tb, :T * tb])
        for (gi, gj), blk in tiles.items():
            want = ref[gi * tb:(gi + 1) * tb, gj * tb:(gj + 1) * tb]
            got = np.tril(blk) if gi == gj else blk
            ok = ok and bool(np.all(np.isfinite(got))) and float(np.max(np.abs(got - want))) < 1e-9
        ok = ok and all(mine(gi, gj) for gi, gj in tiles) and len(tiles) == sum(
            1 for gi in range(T) for gj in range(gi + 1) if mine(gi, gj))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("pr,pc", [(2, 1), (1, 2), (2, 2), (3, 1), (2, 3), (2, 4)])
def test_distributed_cholesky_model_gloo(pr, pc):
    """The algorithm of csrc/dist.hip on gloo ranks: every rank keeps ONLY its tiles of the 2-D block-cyclic
    distribution (a rank that read a tile it does not own would raise KeyError) and must end with exactly its share
    of the factor -- fresh factorisation, then two block appends whose boundaries fall inside a block."""
    world = pr * pc
    gloo_port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_model_worker, args=(r, world, pr, pc, gloo_port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_staircase_tile_enumeration_of_the_distributed_update():
    """The trailing update of a rank enumerates its LOCAL tiles on or below the global diagonal -- a staircase --
    densely (every valid tile exactly once, nothing else), for any grid, block size, region: host replay of
    gemm.hip's map_tile_dense / stair_decode through the C ABI (no GPU needed) against brute force."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _hooks
    cap = 60000
    out = (C.c_int32 * (2 * cap))()
    n_cases = 0
    for pr, pc in [(1, 1), (2, 1), (1, 2), (2, 2), (2, 4), (4, 2), (3, 1), (8, 1), (1, 8), (3, 2)]:
        for nbt in (1, 4):
            for T in (1, 5, 13, 37):
                for rlo, clo in [(0, 0), (3, 3), (7, 2), (T // 2, 1)]:
                    if rlo > T or clo > T:
                        continue
                    for r in range(pr):
                        for c in range(pc):
                            n = _hooks.lib.lpgp_test_stair_enumerate(pr, pc, r, c, nbt, T, rlo, clo, out, cap)
                            assert 0 <= n <= cap
                            got = [tuple(x) for x in np.frombuffer(out, dtype=np.int32, count=2 * n).reshape(n, 2)]
                            want = {(i, j) for i in range(rlo, T) for j in range(clo, T)
                                    if (i // nbt) % pr == r and (j // nbt) % pc == c and i >= j}
                            assert len(got) == len(set(got)) and set(got) == want, (pr, pc, r, c, nbt, T, rlo, clo)
                            n_cases += 1
    assert n_cases > 1000


def test_predict_sharding_bounds():
    import numpy as np
    for n, world in ((4096, 8), (10, 3), (7, 7)):
        b = np.linspace(0, n, world + 1).astype(int)
        assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0) and np.diff(b).max() - np.diff(b).min() <= 1


def _bcast_from_worker(rank, world, port, q):
    import os, sys
    sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd", "linpde_gp_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dist", os.path.join(ROOT, "linpde-gp_amd", "linpde_gp_amd", "_dist.py"))
    _dist = importlib.util.module_from_spec(spec); spec.loader.exec_module(_dist)
    comm = _dist.Comm(rank, world, "127.0.0.1", port)
    got = [comm.bcast_from(("payload", root, bytes(range(7))) if rank == root else None, root) for root in range(world)]
    got.append(comm.allreduce_max(float(rank)))
    comm.barrier()
    comm.close()
    q.put((rank, got))


def test_control_plane_bcast_from_any_root():
    """`Comm.bcast_from` (used by the host-staged test transport of the distributed factorisation)."""
    import multiprocessing as mp
    ctxm = mp.get_context("spawn")
    world, port = 3, 29891
    q = ctxm.Queue()
    ps = [ctxm.Process(target=_bcast_from_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
    for r in range(world):
        assert res[r][:world] == [("payload", root, bytes(range(7))) for root in range(world)]
        assert res[r][world] == float(world - 1)


# ---- wire format and authentication of the control plane (no pickle, authenticate first) ----
def _load_dist():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dist_mod", os.path.join(ROOT, "linpde-gp_amd", "linpde_gp_amd", "_dist.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_wire_format_round_trip_and_rejects_garbage():
    import pickle
    import numpy as np
    _dist = _load_dist()
    msg = (True, None, -7, 2.5, "uid \u00e9", b"\x00\x01\xff", [1, [2.0, (3, "x")]], {"a": np.arange(6.0).reshape(2, 3), 4: np.array([1, 2], dtype=np.int64)},
           np.zeros((0, 2)))
    back = _dist.loads(_dist.dumps(msg))
    assert back[:7] == msg[:7] and isinstance(back[6], list) and isinstance(back[6][1][1], tuple)
    np.testing.assert_array_equal(back[7]["a"], msg[7]["a"])
    np.testing.assert_array_equal(back[7][4], msg[7][4])
    assert back[8].shape == (0, 2)
    for bad in (pickle.dumps(("x", 1)), b"", b"i\x00", b"a" + b"z" + b"\x01", b"l" + (2**40).to_bytes(8, "big"), _dist.dumps(1) + b"N"):
        with pytest.raises(ValueError):
            _dist.loads(bad)
    with pytest.raises(TypeError):
        _dist.dumps(object())
    with pytest.raises(TypeError):
        _dist.dumps(np.zeros(2, dtype=np.float32))


def _auth_worker(rank, world, port, q, key):
    _dist = _load_dist()
    comm = _dist.Comm(rank, world, "127.0.0.1", port, key=key)
    out = comm.allgather(("r", rank))
    ex = comm.exchange({(rank + 1) % world: bytes([rank]) * 3, rank: b"self"})
    comm.close()
    q.put((rank, out, sorted(ex.items())))


def test_control_plane_rejects_unauthenticated_peers_and_exchanges_point_to_point():
    """A connection that does not know the job key (wrong key, a pickle-speaking client of the old protocol,
    raw garbage, an oversized length header) never becomes a peer and nothing it sends is deserialised; the
    job still forms.  Also covers `Comm.exchange` (point-to-point messages of the 2-D test transport)."""
    import pickle
    import struct
    import threading
    import time
    _dist = _load_dist()
    port = _free_port()
    key = b"k" * 32
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_auth_worker, args=(r, world, port, q, key)) for r in range(world)]
    procs[0].start()
    # intruders connect to rank 0 before the real peers do
    def intruder(payload):
        deadline = time.time() + 20
        while time.time() < deadline:
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                break
            except OSError:
                time.sleep(0.05)
        else:
            return
        try:
            s.settimeout(5.0)
            s.sendall(payload)
            s.recv(64)
        except OSError:
            pass
        finally:
            s.close()
    blob = pickle.dumps((("lpgp-comm", world), 1))
    threads = [threading.Thread(target=intruder, args=(p,)) for p in (
        struct.pack("!Q", len(blob)) + blob,                 # the old pickle handshake
        struct.pack("!Q", 1 << 62) + b"x" * 64,              # oversized length header
        struct.pack("!q", 1) + b"n" * 16 + b"h" * 32,        # right shape, wrong key
    )]
    for t in threads:
        t.start()
    # a well-formed client with the WRONG key is refused by both sides
    with pytest.raises((ConnectionError, OSError)):
        s = socket.create_connection(("127.0.0.1", port), timeout=5.0) if any(
            _wait_port(port) for _ in range(1)) else None
        s.settimeout(5.0)
        try:
            _dist._handshake_client(s, b"w" * 32, 1, world)
        finally:
            s.close()
    for p in procs[1:]:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for t in threads:
        t.join(timeout=30)
    for rank, out, ex in res:
        assert out == [("r", 0), ("r", 1), ("r", 2)]
        assert ex == sorted([((rank - 1) % world, bytes([(rank - 1) % world]) * 3), (rank, b"self")])


def _wait_port(port, timeout=20.0):
    import time
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            socket.create_connection(("127.0.0.1", port), timeout=1.0).close()
            return True
        except OSError:
            time.sleep(0.05)
    return False


def test_wire_format_round_trip_property():
    """Any tree of the carried types survives dumps / loads unchanged, and no prefix / corruption of a valid message
    makes `loads` do anything but return or raise ValueError (hypothesis)."""
    import numpy as np
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp
    _dist = _load_dist()
    leaves = st.one_of(st.none(), st.booleans(), st.integers(-2**63, 2**63 - 1), st.floats(allow_nan=False), st.text(max_size=20),
                       st.binary(max_size=40),
                       hnp.arrays(np.float64, hnp.array_shapes(max_dims=3, max_side=4), elements=st.floats(-1e6, 1e6)),
                       hnp.arrays(np.int64, hnp.array_shapes(max_dims=2, max_side=4), elements=st.integers(-10**9, 10**9)))
    trees = st.recursive(leaves, lambda ch: st.one_of(st.lists(ch, max_size=4), st.lists(ch, max_size=4).map(tuple),
                                                      st.dictionaries(st.one_of(st.text(max_size=5), st.integers(-100, 100)), ch, max_size=3)),
                         max_leaves=12)

    def same(a, b):
        if isinstance(a, np.ndarray):
            return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
        if isinstance(a, (list, tuple)):
            return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, dict):
            return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        return type(a) is type(b) and a == b

    @settings(max_examples=150, deadline=None)
    @given(trees, st.data())
    def check(tree, data):
        blob = _dist.dumps(tree)
        assert same(_dist.loads(blob), tree)
        cut = data.draw(st.integers(0, len(blob)))
        flip = data.draw(st.integers(0, max(len(blob) - 1, 0)))
        for bad in (blob[:cut], blob[:flip] + bytes([blob[flip] ^ 0x5A]) + blob[flip + 1:] if blob else b""):
            try:
                _dist.loads(bad)
            except (ValueError, UnicodeDecodeError):
                pass

    check()
