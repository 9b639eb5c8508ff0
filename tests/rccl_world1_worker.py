"""Worker of tests/test_rccl_world1.py, started by `python -m torch.distributed.run --nproc-per-node 1` on the GPU box: the N > 1
structure of bench.py at world size 1 on real hardware — init_process_group("nccl") (= RCCL), the HIP kernels writing a device slab
[keypoints | descriptors | n | mono], an asynchronous gather of that slab to rank 0 double-buffered against the next step's compute,
unpack_slab of what arrived, compared with the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import extractorb_amd as X                      # noqa: E402
from extractorb_amd import sharding, synth      # noqa: E402
import oracle_lib as O                          # noqa: E402  (the checker)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    B, rows, cols, nf, steps = 6, 480, 640, 1000, 4
    ex = X.ORBextractor(nf, 1.2, 8, 20, 7, max_width=cols, max_height=rows, max_batch=B, device=local)
    ex.set_stream(torch.cuda.current_stream().cuda_stream)
    cap = nf + 3 * 8
    lay = sharding.slab_layout(B, cap)
    slabs = [torch.zeros(lay["bytes"], dtype=torch.uint8, device="cuda") for _ in range(2)]
    gathered = [[torch.empty_like(slabs[0]) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
    pending = [None, None]

    def frames_of(step, r):
        """rank r's frames of a step: a pure function of (step, rank), so that rank 0 can rebuild what every other rank extracted"""
        lo, hi = sharding.shard_range(world * B, r, world)
        return synth.frames(["textured", "noise"][step & 1], 1000 * step + lo, hi - lo, rows, cols)

    for s in range(steps):
        fr = frames_of(s, rank)
        k = s & 1
        if pending[k] is not None:
            pending[k].wait()
            if rank == 0:       # the slab of step s - 2 has arrived: check it before its buffers are reused
                check(gathered[k], lambda r: frames_of(s - 2, r), B, cap, nf, world)
        b = slabs[k].data_ptr()
        ex.extract_batch_device(torch.from_numpy(fr).cuda(), B, rows, cols, b + lay["keypoints"], b + lay["descriptors"], b + lay["n"],
                                b + lay["mono"], cap)
        pending[k] = dist.gather(slabs[k], gathered[k] if rank == 0 else None, dst=0, async_op=True)
    for s in (steps - 2, steps - 1):
        pending[s & 1].wait()
        torch.cuda.synchronize()
        if rank == 0:
            check(gathered[s & 1], lambda r, s=s: frames_of(s, r), B, cap, nf, world)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_WORLD1_OK steps=%d frames_per_step=%d" % (steps, B))


def check(bufs, frames_of_rank, B, cap, nf, world):
    """every rank's gathered slab against the oracle on THAT rank's frames"""
    torch.cuda.synchronize()
    o = O.Oracle(nf)
    for r in range(world):
        got = sharding.unpack_slab(bufs[r].cpu().numpy(), B, cap)
        frames = frames_of_rank(r)
        for f in range(B):
            mono, k, d = o.extract(frames[f])
            assert got[f][0] == mono and got[f][1].tobytes() == k.tobytes() and np.array_equal(got[f][2], d), "rank %d frame %d" % (r, f)


if __name__ == "__main__":
    main()
