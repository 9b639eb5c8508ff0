"""Frame sharding across ranks and the gather of per-frame result slabs.

Frames are independent units: rank r of W owns a contiguous block of the stream and runs the whole path on
it with no data-path collective.  The only communication is moving finished result slabs to rank 0
(`torch.distributed.gather`: RCCL over xGMI on the GPUs, gloo in the CPU tests).
"""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one frame and cover the stream."""
    if world < 1 or not (0 <= rank < world) or n_frames < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def slab_layout(frames_per_rank, capacity):
    """Byte offsets inside one rank's result slab: [keypoints | descriptors | n | mono], 256-byte padded."""
    off_k = 0
    off_d = off_k + frames_per_rank * capacity * 28
    off_n = off_d + frames_per_rank * capacity * 32
    off_m = off_n + 4 * frames_per_rank
    total = (off_m + 4 * frames_per_rank + 255) // 256 * 256
    return dict(keypoints=off_k, descriptors=off_d, n=off_n, mono=off_m, bytes=total)


def pack_slab(results, frames_per_rank, capacity):
    """results: list of (mono, keypoints[structured 28 B], descriptors[n,32]) -> uint8 slab (numpy)."""
    lay = slab_layout(frames_per_rank, capacity)
    slab = np.zeros(lay["bytes"], np.uint8)
    n_arr = slab[lay["n"]:lay["n"] + 4 * frames_per_rank].view(np.int32)
    m_arr = slab[lay["mono"]:lay["mono"] + 4 * frames_per_rank].view(np.int32)
    for f, (mono, k, d) in enumerate(results):
        n = len(k)
        if n > capacity:
            raise ValueError("frame %d: %d keypoints exceed the slab capacity %d" % (f, n, capacity))
        ko = lay["keypoints"] + f * capacity * 28
        do = lay["descriptors"] + f * capacity * 32
        slab[ko:ko + n * 28] = np.frombuffer(k.tobytes(), np.uint8)
        slab[do:do + n * 32] = d.reshape(-1)
        n_arr[f], m_arr[f] = n, mono
    return slab


def unpack_slab(slab, frames_per_rank, capacity, n_valid=None, keypoint_dtype=None):
    """Inverse of pack_slab (also decodes slabs written by the HIP kernels, whose layout is the same)."""
    from .orbextractor import KEYPOINT_DTYPE
    dt = keypoint_dtype or KEYPOINT_DTYPE
    slab = np.asarray(slab, np.uint8)
    lay = slab_layout(frames_per_rank, capacity)
    n_arr = slab[lay["n"]:lay["n"] + 4 * frames_per_rank].view(np.int32)
    m_arr = slab[lay["mono"]:lay["mono"] + 4 * frames_per_rank].view(np.int32)
    out = []
    for f in range(frames_per_rank if n_valid is None else n_valid):
        n = int(n_arr[f])
        if not 0 <= n <= capacity:
            raise ValueError("frame %d: keypoint count %d outside the slab capacity %d (corrupt or mismatched slab)" % (f, n, capacity))
        ko = lay["keypoints"] + f * capacity * 28
        do = lay["descriptors"] + f * capacity * 32
        k = slab[ko:ko + n * 28].copy().view(dt).reshape(-1)
        d = slab[do:do + n * 32].copy().reshape(n, 32)
        out.append((int(m_arr[f]), k, d))
    return out


def gather_slabs(slab_tensor, dst=0):
    """Gathers every rank's slab tensor to `dst`; returns the list there, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(slab_tensor) for _ in range(world)] if rank == dst else None
    dist.gather(slab_tensor, bufs, dst=dst)
    return bufs
