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


def test_one_rank_nccl_group_carries_the_real_exchange_buffers():
    """RCCL on the hardware there is.  A 1-rank `nccl` process group on the 1-GPU box, and MIPS / ShardedSearcher
    forced to take the multi-rank path through it (`force_collectives`): the aux-layout and sample all-gathers (int32), the packed
    record all-gather (uint8), the return_idxs SUM all-reduce (fp32), results = the plain single-rank path (tests/_nccl_one_rank_worker.py)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(here, "_nccl_one_rank_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
