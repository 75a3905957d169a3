"""Multi-GPU host logic of the row-tiled render: one process per
GPU, the output rows split into contiguous strips, the source coefficients
replicated by ONE broadcast per source, strips gathered on rank 0 at the end.
No collective runs inside a render step. The functions take torch tensors on
whatever device the process group's backend serves (RCCL: device tensors over
xGMI; gloo: CPU tensors, used by the CPU tests)."""


def row_partition(height, world_size, rank, align=1):
    """rows [r0, r1) of `rank`: contiguous, covering, sizes differ by < align+1"""
    if not (0 <= rank < world_size):
        raise ValueError("rank outside the group")
    units = (height + align - 1) // align
    u0 = (units * rank) // world_size
    u1 = (units * (rank + 1)) // world_size
    return min(u0 * align, height), min(u1 * align, height)


def broadcast_source(dist, tensor, src_rank=0):
    """replicate a coefficient container (flat float32 tensor, already allocated
    with the same size on every rank) from src_rank"""
    dist.broadcast(tensor, src_rank)
    return tensor


def gather_strips(dist, strip, height, width, nch, rank, world_size, dst=0, align=1):
    """collect every rank's (rows, width, nch) strip on `dst`; returns the full
    frame there, None elsewhere. Strips may differ in height: point-to-point
    sends into views of the frame (over xGMI every sender uses its own link)."""
    import torch
    if world_size == 1:
        return strip
    if rank == dst:
        frame = torch.empty((height, width, nch), dtype=strip.dtype, device=strip.device)
        r0, r1 = row_partition(height, world_size, rank, align)
        frame[r0:r1].copy_(strip)
        reqs = []
        for r in range(world_size):
            if r == dst:
                continue
            q0, q1 = row_partition(height, world_size, r, align)
            if q1 > q0:
                reqs.append(dist.irecv(frame[q0:q1], src=r))
        for q in reqs:
            q.wait()
        return frame
    if strip.shape[0] > 0:
        dist.send(strip.contiguous(), dst=dst)
    return None
