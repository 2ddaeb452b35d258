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
