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


def _exchange(dist, ops):
    """post all point-to-point operations of a gather as ONE batch (RCCL runs them as a group:
    every sender's link is busy at once) and wait; backends without batching take them one by one"""
    if not ops:
        return
    try:
        reqs = dist.batch_isend_irecv([dist.P2POp(fn, t, peer) for fn, t, peer in ops])
    except (RuntimeError, AttributeError, NotImplementedError):
        reqs = [fn(t, peer) for fn, t, peer in ops]
    for q in reqs:
        q.wait()


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
        ops = []
        for r in range(world_size):
            if r == dst:
                continue
            q0, q1 = row_partition(height, world_size, r, align)
            if q1 > q0:
                ops.append((dist.irecv, frame[q0:q1], r))
        _exchange(dist, ops)
        return frame
    if strip.shape[0] > 0:
        _exchange(dist, [(dist.isend, strip.contiguous(), dst)])
    return None


# ---- interleaved row bands (eu_target.band_*) -----------------------------------
# Rows of a frame can differ in cost (the polar faces of a cubemap target take
# 1.7x the others: contiguous strips leave 8 GPUs at 5.6x, tools/strip_times.py).
# Dealing bands of rows round-robin gives every rank a slice of every region.

def band_frame_rows(height, band_rows, world_size, rank):
    """frame rows of `rank`, in the order of its local rows (torch int64, CPU)"""
    import torch
    y = torch.arange(height, dtype=torch.int64)
    if world_size <= 1:
        return y
    return y[(y // band_rows) % world_size == rank]


def band_local_rows(height, band_rows, world_size, rank):
    return int(band_frame_rows(height, band_rows, world_size, rank).numel())


def gather_bands(dist, strip, height, width, nch, rank, world_size, band_rows, dst=0):
    """collect every rank's compacted (local_rows, width, nch) band set on `dst`
    and put the rows where they belong; returns the frame there, None elsewhere"""
    import torch
    if world_size == 1:
        return strip
    if rank == dst:
        frame = torch.empty((height, width, nch), dtype=strip.dtype, device=strip.device)
        frame[band_frame_rows(height, band_rows, world_size, rank).to(strip.device)] = strip
        bufs, ops = {}, []
        for r in range(world_size):
            if r == dst:
                continue
            n = band_local_rows(height, band_rows, world_size, r)
            if n:
                bufs[r] = torch.empty((n, width, nch), dtype=strip.dtype, device=strip.device)
                ops.append((dist.irecv, bufs[r], r))
        _exchange(dist, ops)
        for r, b in bufs.items():
            frame[band_frame_rows(height, band_rows, world_size, r).to(strip.device)] = b
        return frame
    if strip.shape[0] > 0:
        _exchange(dist, [(dist.isend, strip.contiguous(), dst)])
    return None


# ---- contiguous strips of equal estimated cost --------------------------------------
# Where the library can say which rows are the expensive ones (eu_hip_layout_segments:
# the segments it renders with the tile layout: 1.55x the time of the others, 2x counting
# the extra launches they break a strip into - measured, tools/strip_times.py) a
# frame is better cut into CONTIGUOUS strips of equal cost than dealt out in bands: every
# rank then has a few long runs of one layout and the launch-level layout choice applies
# to it (it does not for the many short runs of a band-interleaved share).

def cost_partition(height, world_size, seg_rows, flags, flag_cost=2.0, align=64):
    """[(r0, r1)] for every rank: contiguous, covering, boundaries at multiples of
    `align`, cumulative cost (1 per row, flag_cost per row of a flagged segment) split
    evenly. Deterministic: every rank computes the same list."""
    import numpy as np
    w = np.ones(height, np.float64)
    for k, f in enumerate(np.asarray(flags).tolist()):
        if f:
            w[k * seg_rows:(k + 1) * seg_rows] = flag_cost
    cum = np.concatenate([[0.0], np.cumsum(w)])
    if height < world_size * align:
        # too small to give every rank an aligned strip by cost: plain equal strips (a rank may
        # still come out empty when there are fewer rows than ranks; callers skip its render)
        return [row_partition(height, world_size, r) for r in range(world_size)]
    bounds = [0]
    for r in range(1, world_size):
        y = int(np.searchsorted(cum, cum[-1] * r / world_size))
        y = int(round(y / align)) * align
        # every rank keeps at least `align` rows, also the ones behind this boundary
        y = min(max(y, bounds[-1] + align), height - (world_size - r) * align)
        bounds.append(y)
    bounds.append(height)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def gather_ranges(dist, strip, ranges, height, width, nch, rank, world_size, dst=0):
    """collect every rank's rows [ranges[r][0], ranges[r][1]) on `dst`; returns the frame
    there, None elsewhere"""
    import torch
    if world_size == 1:
        return strip
    if rank == dst:
        frame = torch.empty((height, width, nch), dtype=strip.dtype, device=strip.device)
        r0, r1 = ranges[rank]
        frame[r0:r1].copy_(strip)
        ops = []
        for r in range(world_size):
            q0, q1 = ranges[r]
            if r != dst and q1 > q0:
                ops.append((dist.irecv, frame[q0:q1], r))
        _exchange(dist, ops)
        return frame
    if strip.shape[0] > 0:
        _exchange(dist, [(dist.isend, strip.contiguous(), dst)])
    return None
