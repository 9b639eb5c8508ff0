"""The RCCL leg of the multi-GPU path on real hardware at world size 1 (the driver owns the 8-GPU runs): the launcher starts a
rank before anything touches the GPU, the rank initialises the nccl backend, the HIP kernels fill a device slab, the slab is
gathered asynchronously and decoded with sharding.unpack_slab, and every frame is compared with the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_device_slab_gather_over_rccl_at_world_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_world1_worker.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
