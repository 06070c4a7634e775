"""The sharded product path across real process boundaries on ONE GPU: 2 and 3 processes share cuda:0, every process
holds its own range shard in its own libdph handle and the exchanges go over gloo (tests/_gloo_gpu_worker.py).  What the
RCCL test (test_dist_nccl.py) needs two GPUs for, minus the RCCL transport itself."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,ivf", [(2, 0), (3, 0), (4, 0), (8, 0), (4, 1)])
def test_mips_range_sharded_over_processes_sharing_one_gpu(world, ivf):
    """world 4 and 8: the rank counts of configs[3] and configs[2] (no 4- or 8-GPU node is available to the builder: the protocol is
    rehearsed with that many PROCESSES on the one GPU); ivf = 1: the shards stored list-major, IVF search (configs[3])."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DPH_GLOO_WORKER_IVF"] = str(ivf)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_gloo_gpu_worker.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
