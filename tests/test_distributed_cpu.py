"""The N>1 host path on CPU: two gloo ranks broadcast a source container,
render their row strips (with the oracle standing in for the GPU kernel - this
is a test of the host logic, not of the kernel) and gather the frame."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_partition_covers_without_overlap():
    from envutil_amd.distributed import row_partition
    for h in (1, 7, 100, 24576, 24577):
        for w in (1, 2, 3, 4, 8):
            for align in (1, 4):
                rows = [row_partition(h, w, r, align) for r in range(w)]
                assert rows[0][0] == 0 and rows[-1][1] == h
                for a, b in zip(rows, rows[1:]):
                    assert a[1] == b[0] and a[0] <= a[1]
                if align == 1:
                    sizes = [b - a for a, b in rows]
                    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        row_partition(10, 2, 2)


def test_band_rows_cover_without_overlap():
    """interleaved bands: every frame row belongs to exactly one part; the
    library's count (eu_hip_band_rows) agrees with the host-side row lists"""
    import envutil_amd as ea
    from envutil_amd.distributed import band_frame_rows, band_local_rows
    for h in (1, 7, 100, 192, 24576, 24577, 1000):
        for w in (1, 2, 3, 4, 8):
            for br in (4, 8, 64):
                parts = [band_frame_rows(h, br, w, r).numpy() for r in range(w)]
                allrows = np.sort(np.concatenate(parts))
                assert np.array_equal(allrows, np.arange(h))
                for r in range(w):
                    assert len(parts[r]) == ea.band_rows(h, br, w, r) == band_local_rows(h, br, w, r)
                    assert np.array_equal(parts[r], ea.band_frame_rows(h, br, w, r))
                    # local row l -> frame row, the kernel's formula (eu_device.h: eu_frame_row)
                    l = np.arange(len(parts[r]))
                    assert np.array_equal(((l // br) * w + r) * br + l % br, parts[r])
                if w > 1 and h >= br * w:
                    sizes = [len(p) for p in parts]
                    assert max(sizes) - min(sizes) <= br


def test_cost_partition_covers_and_balances():
    """contiguous strips of equal estimated cost (distributed.cost_partition)"""
    from envutil_amd.distributed import cost_partition
    flags = np.zeros(48, np.uint8)
    flags[18:22] = 1
    flags[26:30] = 1                                  # the headline job's flagged segments
    for world in (1, 2, 3, 4, 8):
        rg = cost_partition(24576, world, 512, flags)
        assert rg[0][0] == 0 and rg[-1][1] == 24576 and len(rg) == world
        cost = []
        for (a, b), nxt in zip(rg, rg[1:] + [(24576, 24576)]):
            assert a <= b and b == nxt[0] and a % 64 == 0
            w = np.ones(24576)
            for k in np.flatnonzero(flags):
                w[k * 512:(k + 1) * 512] = 2.0
            cost.append(w[a:b].sum())
        assert max(cost) - min(cost) <= 2 * 64 * 2.0 + 1e-9, cost
    # no flags: equal row counts; odd heights
    rg = cost_partition(1000, 3, 512, np.zeros(2, np.uint8))
    assert rg[0][0] == 0 and rg[-1][1] == 1000 and all(a <= b for a, b in rg)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import envutil_amd as ea
    from envutil_amd.distributed import row_partition, broadcast_source, gather_strips
    import euo
    import jobs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img = jobs.synth_image(128, 64, 3)
        # rank 0 owns the prefiltered coefficients; rank 1 starts from garbage
        o = jobs.OracleSource(euo.SPHERICAL, 128, 64, 360.0, img if rank == 0 else img * 0 + 7, 3)
        flat = torch.from_numpy(o.container.reshape(-1))
        broadcast_source(dist, flat, 0)             # in place: o.container now rank 0's
        a = ea.arguments(ea.CUBEMAP, 32, 192, 90.0, spline_degree=3)
        r0, r1 = row_partition(a.height, world, rank)
        strip = torch.from_numpy(jobs.oracle_render(a, o, row_begin=r0, row_end=r1, nthreads=2))
        frame = gather_strips(dist, strip, a.height, a.width, 3, rank, world)
        # the same frame dealt out in interleaved bands of 8 rows (eu_target.band_*)
        from envutil_amd.distributed import band_frame_rows, gather_bands
        rows = band_frame_rows(a.height, 8, world, rank).numpy()
        starts = rows[::8]
        bands = [jobs.oracle_render(a, o, row_begin=int(s), row_end=int(min(s + 8, a.height)), nthreads=2)
                 for s in starts]
        mine = torch.from_numpy(np.concatenate(bands))
        assert mine.shape[0] == ea.band_rows(a.height, 8, world, rank)
        frame2 = gather_bands(dist, mine, a.height, a.width, 3, rank, world, 8)
        # and as contiguous strips of equal estimated cost (an arbitrary flag pattern)
        from envutil_amd.distributed import cost_partition, gather_ranges
        ranges = cost_partition(a.height, world, 16, np.array([0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0], np.uint8), align=4)
        c0, c1 = ranges[rank]
        mine3 = torch.from_numpy(jobs.oracle_render(a, o, row_begin=c0, row_end=c1, nthreads=2))
        frame3 = gather_ranges(dist, mine3, ranges, a.height, a.width, 3, rank, world)
        if rank == 0:
            whole = jobs.oracle_render(a, jobs.OracleSource(euo.SPHERICAL, 128, 64, 360.0, img, 3), nthreads=2)
            q.put(bool((frame.numpy().view(np.uint32) == whole.view(np.uint32)).all()) and
                  bool((frame2.numpy().view(np.uint32) == whole.view(np.uint32)).all()) and
                  bool((frame3.numpy().view(np.uint32) == whole.view(np.uint32)).all()))
        else:
            q.put(frame is None and frame2 is None and frame3 is None)
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_render_gather_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [True, True]
