"""The sharded product path across real process boundaries on ONE GPU: 2 and 3 processes share cuda:0, every process
holds its own range shard in its own libdph handle and the exchanges go over gloo (tests/_gloo_gpu_worker.py).  What the
RCCL test (test_dist_nccl.py) needs two GPUs for, minus the RCCL transport itself."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_mips_range_sharded_over_processes_sharing_one_gpu(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_gloo_gpu_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
