"""Range-sharded MIPS over RCCL, one process per GPU (needs >= 2 GPUs; the 1-GPU box skips it -- the same code path is
exercised there with thread ranks in test_gpu_search.py::test_mips_range_sharded_over_ranks_equals_single_rank, and the
exchange logic on CPU with gloo in test_dist_gloo.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2])
def test_mips_sharded_over_rccl(world):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_nccl_worker.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
