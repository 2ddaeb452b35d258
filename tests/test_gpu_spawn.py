"""The single-process front of the multi-GPU path (`linpde_gp_amd.spawn`, SURVEY.md §8b "one process drives all 8 GPUs"):
an unmodified reference-style script -- one process, plain `condition_on_observations` / `predict` calls -- whose
posterior is built by worker processes, one per rank.  The test box has ONE GPU, so the two workers share it: through the
product's RCCL code path over loopback sockets (distinct NCCL_HOSTID per rank) and through the direct-peer IPC transport.
Posterior vs the CPU oracle with the one criterion of tests/conftest.py; API errors re-raised in the calling process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("transport,ranks", [("rccl", 2), ("ipc", 3)])
def test_single_process_front(transport, ranks):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LPGP_IPC_WINDOW_MB="8", OPENBLAS_NUM_THREADS="8", OMP_NUM_THREADS="8")
    env.pop("LPGP_SPAWN", None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_spawn_script.py"), transport, str(ranks)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "SPAWN-OK" in out.stdout, out.stdout[-2000:] + "\n" + out.stderr[-4000:]
    print(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("mode", ["call", "env"])
def test_unguarded_script_and_lpgp_spawn_env(mode):
    """workers run `python -m linpde_gp_amd._spawn_worker`, never the caller's script -- so a script without an
    `if __name__ == "__main__":` guard works, both with an explicit `lp.spawn(2)` and with nothing but LPGP_SPAWN=2 in the
    environment (with multiprocessing's "spawn" start method every worker re-ran the script and called spawn() again)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LPGP_IPC_WINDOW_MB="8", OPENBLAS_NUM_THREADS="8", OMP_NUM_THREADS="8")
    env.pop("LPGP_SPAWN", None)
    if mode == "env":
        env.update(LPGP_SPAWN="2", LPGP_SPAWN_DEVICES="0,0", LPGP_SPAWN_TRANSPORT="ipc")
    out = subprocess.run([sys.executable, os.path.join(HERE, "_spawn_unguarded.py"), mode], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "SPAWN-OK" in out.stdout, out.stdout[-2000:] + "\n" + out.stderr[-4000:]
