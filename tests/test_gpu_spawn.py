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
