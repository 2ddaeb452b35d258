"""Entry module of a worker process of the single-process multi-GPU front (`_spawn.WorkerGroup`).

    python -m linpde_gp_amd._spawn_worker <address of the parent's listener> <rank>

A fresh interpreter that runs THIS module -- the caller's script is never imported here -- dials back to the parent over
the authenticated AF_UNIX connection (key in $LPGP_SPAWN_AUTHKEY, removed from the environment at once), receives its
assignment and enters the request loop of `_spawn._worker_main`.  No GPU call happens before the assignment has set the
environment of the rank.
"""
import os
import sys
from multiprocessing.connection import Client


def main(argv):
    address, rank = argv[1], int(argv[2])
    key = bytes.fromhex(os.environ.pop("LPGP_SPAWN_AUTHKEY"))
    os.environ.pop("LPGP_SPAWN", None)
    conn = Client(address, family="AF_UNIX", authkey=key)
    conn.send(rank)
    r, world, port, transport, device, grid, extra_env = conn.recv()
    from linpde_gp_amd import _spawn
    _spawn._worker_main(r, world, port, conn, transport, device, grid, extra_env)


if __name__ == "__main__":
    main(sys.argv)
