"""The N>1 path on CPU: two gloo ranks shard a frame stream, each extracts its block, and the per-frame result
slabs are gathered to rank 0 — exactly the structure bench.py runs over RCCL.  The per-rank extractor here is the
CPU oracle (no GPU in this container); the sharding, slab layout and gather code are the product's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from extractorb_amd import sharding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES, ROWS, COLS, NF = 5, 240, 320, 300     # odd frame count: ranks get 3 and 2 frames
CAP = NF + 3 * 8 + 64


def _worker(rank, world, port, out_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(N_FRAMES, rank, world)
    per_rank = (N_FRAMES + world - 1) // world
    frames = synth.frames("textured", lo, hi - lo, ROWS, COLS)
    o = O.Oracle(NF)
    results = [o.extract(f) for f in frames]
    slab = torch.from_numpy(sharding.pack_slab(results, per_rank, CAP))
    got = sharding.gather_slabs(slab, dst=0)
    if rank == 0:
        allres = []
        for r in range(world):
            rlo, rhi = sharding.shard_range(N_FRAMES, r, world)
            allres += sharding.unpack_slab(got[r].numpy(), per_rank, CAP, n_valid=rhi - rlo, keypoint_dtype=O.KEYPOINT_DTYPE)
        np.savez(out_path, n=np.array([len(k) for _, k, _ in allres]), mono=np.array([m for m, _, _ in allres]),
                 k=np.concatenate([np.frombuffer(k.tobytes(), np.uint8) for _, k, _ in allres]),
                 d=np.concatenate([d.reshape(-1) for _, _, d in allres]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_the_stream():
    for n in (0, 1, 5, 64, 511, 512):
        for w in (1, 2, 3, 8):
            blocks = [sharding.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_slab_roundtrip():
    import oracle_lib as O
    o = O.Oracle(NF)
    res = [o.extract(f) for f in synth.frames("noise", 0, 2, ROWS, COLS)]
    back = sharding.unpack_slab(sharding.pack_slab(res, 2, CAP), 2, CAP, keypoint_dtype=O.KEYPOINT_DTYPE)
    for a, b in zip(res, back):
        assert a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2])
    with pytest.raises(ValueError):
        sharding.pack_slab(res, 2, 10)


def test_two_rank_gloo_gather_equals_single_process(tmp_path):
    import oracle_lib as O
    out = str(tmp_path / "gathered.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    z = np.load(out)
    o = O.Oracle(NF)
    want = [o.extract(f) for f in synth.frames("textured", 0, N_FRAMES, ROWS, COLS)]
    assert z["n"].tolist() == [len(k) for _, k, _ in want]
    assert z["mono"].tolist() == [m for m, _, _ in want]
    assert z["k"].tobytes() == b"".join(k.tobytes() for _, k, _ in want)
    assert z["d"].tobytes() == b"".join(d.tobytes() for _, _, d in want)
