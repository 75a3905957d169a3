#!/usr/bin/env python3
"""bench.py - whole-job throughput of the MI355X-native reprojection path.

One "step" = one pass of the hot path over one frame: every output pixel of the
headline job (16384x8192 lat/lon -> 6x4096 cubemap, b-spline degree 3 with
prefilter, RGB float32) is produced once, with the prefiltered source
coefficients already resident in HBM. With N GPUs the output rows are tiled
across the ranks (source replicated by one RCCL broadcast at set-up, no
collective inside a step); value = pixels all ranks produced / max-over-ranks
time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

Prints ONE JSON line on rank 0 (see the driver contract in the task
description), including
  roofline     - algorithmic bytes per launch / mean kernel time measured with
                 HIP events on the kernel's own stream, against 8 TB/s HBM
  cpu_baseline - the CPU oracle (a port of the reference path, NOT envutil's
                 SIMD binary) timed on this box's host cores on a band of rows
                 of the same frame.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (source (prj, w, h, hfov), target (prj, w, h, hfov), nch, degree, twine, ypr)
    "headline": (("spherical", 16384, 8192, 360.0), ("cubemap", 4096, 24576, 90.0), 3, 3, 0, (0, 0, 0)),
    # BASELINE config 1 (the reference's own CPU-runnable case: plumbing; 1 Mpix - a test case, not a bench line)
    "config1": (("spherical", 2048, 1024, 360.0), ("rectilinear", 1024, 1024, 90.0), 3, 1, 0, (0, 0, 0)),
    "config2": (("spherical", 8192, 4096, 360.0), ("cubemap", 2048, 12288, 90.0), 3, 1, 0, (0, 0, 0)),
    "config3": (("cubemap", 2048, 12288, 90.0), ("spherical", 16384, 8192, 360.0), 3, 3, 0, (0, 0, 0)),
    "config4": (("spherical", 32768, 16384, 360.0), ("spherical", 32768, 16384, 360.0), 3, 1, 3, (30, 15, 7.5)),
    "small": (("spherical", 2048, 1024, 360.0), ("cubemap", 512, 3072, 90.0), 3, 3, 0, (0, 0, 0)),
    # BASELINE config 5: six 8192^2 RGBA circular-fisheye facets (hfov 130, yaw 0/90/180/270 +
    # pitch +-90, PTO lens a=.01 b=-.03 c=.02) -> 16384x8192 spherical, voronoi_syn_plus
    "config5": (("fisheye", 8192, 8192, 130.0), ("spherical", 16384, 8192, 360.0), 4, 1, 0, (0, 0, 0)),
    # diagnostic shapes (not bench lines): headline-sized target, cache-resident source
    "probe_smallsrc": (("spherical", 2048, 1024, 360.0), ("cubemap", 4096, 24576, 90.0), 3, 3, 0, (0, 0, 0)),
    "probe_bilinear": (("spherical", 16384, 8192, 360.0), ("cubemap", 4096, 24576, 90.0), 3, 1, 0, (0, 0, 0)),
}
WORKLOAD_TEXT = {
    "headline": "16384x8192 lat/lon -> 6x4096 cubemap, b-spline degree 3 + prefilter, RGB f32",
    "config1": "2048x1024 lat/lon -> 1024x1024 rectilinear, hfov 90, bilinear, RGB f32",
    "config2": "8192x4096 lat/lon -> 6x2048 cubemap, bilinear, RGB f32",
    "config3": "6x2048 cubemap -> 16384x8192 spherical, b-spline degree 3 + prefilter, RGB f32",
    "config4": "32768x16384 lat/lon -> 32768x16384 spherical, ypr 30/15/7.5, 3x3 twining, bilinear, RGB f32",
    "small": "2048x1024 lat/lon -> 6x512 cubemap, b-spline degree 3 (smoke size)",
    "config5": "6 x 8192^2 RGBA fisheye facets (hfov 130, lens a/b/c) -> 16384x8192 spherical, bilinear, voronoi_syn_plus",
    "probe_smallsrc": "DIAGNOSTIC 2048x1024 lat/lon -> 6x4096 cubemap, degree 3",
    "probe_bilinear": "DIAGNOSTIC 16384x8192 lat/lon -> 6x4096 cubemap, bilinear",
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
BAND_ROWS = 64          # multi-GPU tiling unit (eu_target.band_rows): two XCD units of the kernels


class _DevBuf:
    """exposes a raw device pointer to torch through __cuda_array_interface__"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4",
                                         "data": (ptr, False), "version": 2}


def synth_on_device(torch, dev, w, h, nch):
    """SURVEY.md 8(d) synthetic field, generated on the GPU (f32)"""
    x = torch.arange(w, device=dev, dtype=torch.float32)[None, :]
    y = torch.arange(h, device=dev, dtype=torch.float32)[:, None]
    img = torch.empty((h, w, nch), device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev)
    for c in range(nch):
        smooth = 0.5 + 0.25 * torch.sin(2 * math.pi * (3 + c) * x / w) * torch.cos(2 * math.pi * (2 + c) * y / h)
        g.manual_seed(12345 + c)
        noise = torch.rand((h, w), device=dev, generator=g)
        img[:, :, c] = smooth + 0.05 * noise - 0.025
    return img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--probe-stages", action="store_true",
                    help="diagnostic: also time the kernel stopped after the ray / coordinate stage")
    a = ap.parse_args()

    import numpy as np
    import torch
    import envutil_amd as ea

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE is {world}: launch with "
                         "torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available() or ea.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    # EU_BENCH_REHEARSE=1: the N-rank logic on however many GPUs there are (all ranks may
    # share one), gloo instead of RCCL, host-staged transfers, and a check of the gathered
    # frame against a single launch - a rehearsal of the multi-GPU path, not a measurement
    rehearse = os.environ.get("EU_BENCH_REHEARSE") == "1"
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    ea.lib().eu_hip_init(local)
    dist = None
    # EU_BENCH_FORCE_DIST=1: take the N-rank code path (RCCL process group, broadcast of the source into
    # the library's buffer, barrier, all-reduce of the time, gather) with ONE rank - a check of the
    # transport plumbing on a one-GPU box, not a measurement
    force_dist = world == 1 and os.environ.get("EU_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1 or force_dist:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=dev)

    (sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = WORKLOADS[a.workload]
    from envutil_amd.api import PROJECTION_NAMES
    sprj, tprj = PROJECTION_NAMES.index(sname), PROJECTION_NAMES.index(tname)

    # ---- set-up (untimed): source coefficients into HBM on every rank ------
    if a.workload == "config5":
        views = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
        fcts = [ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch, yaw=v[0], pitch=v[1], roll=v[2],
                              lens=dict(a=0.01, b=-0.03, c=0.02)) for v in views]
    else:
        fcts = [ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch)]
    t_setup = time.time()
    t_load = 0.0
    sources = []
    for fct in fcts:
        if rank == 0:
            img = synth_on_device(torch, dev, sw, sh, nch)
            if nch in (2, 4):
                img[:, :, nch - 1] = 1.0
            host = img.cpu().numpy()
            del img
            torch.cuda.empty_cache()
            tl = time.perf_counter()
            s1 = ea.Source.load(fct, host, degree)       # H2D + device prefilter/brace
            t_load += time.perf_counter() - tl
            del host
        else:
            s1 = ea.Source.alloc(fct, degree)
        if dist is not None:
            ptr, n = s1.device_ptr()
            buf = torch.as_tensor(_DevBuf(ptr, n), device=dev)
            if rehearse:
                hb = buf.cpu()
                dist.broadcast(hb, 0)
                if rank != 0:
                    buf.copy_(hb)
                del hb
            else:
                dist.broadcast(buf, 0)                   # RCCL over xGMI, once per source
            torch.cuda.synchronize()
        sources.append(s1)
    src = sources[0]
    t_setup = time.time() - t_setup

    args = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2],
                        spline_degree=degree, twine=twine)
    # this rank's rows: interleaved bands of BAND_ROWS rows dealt round-robin
    # (rows differ in cost - the polar cube faces take 1.7x the others - so
    # contiguous strips would leave 8 GPUs at 5.6x, tools/strip_times.py)
    # ... unless the library can say which rows are the expensive ones (lat/lon source,
    # cubic taps: the segments it renders with the tile layout): then contiguous strips
    # of equal estimated cost, on which the launch-level layout choice keeps working
    from envutil_amd.distributed import gather_bands, gather_ranges, cost_partition
    band = (BAND_ROWS, world, rank) if world > 1 else None
    r0, r1 = 0, ea.band_rows(th, BAND_ROWS, world, rank) if world > 1 else th
    ranges = None
    if world > 1 and degree >= 2 and not twine and len(sources) == 1:
        seg_rows, flags = ea.layout_segments(args, sources, nch)
        if flags.size and flags.any():
            ranges = cost_partition(th, world, seg_rows, flags)
            band = None
            r0, r1 = ranges[rank]
    out = torch.empty(((r1 - r0), tw, nch), device=dev, dtype=torch.float32)
    srcs = (C.c_void_p * len(sources))(*[x.handle for x in sources])
    nsrcs = len(sources)
    tgt = args.target(nch, r0, r1, 0, band)

    # the steps run on a stream of their own, so that HIP events on THAT stream bracket the timed steps
    # (torch.cuda.Event sees only the stream it is recorded on)
    st = torch.cuda.Stream(device=dev)
    # (its hardware queue is made on first use, which takes ~5 ms of host time: left to the first warm-up step that
    # is 5 ms of idle GPU right behind the clock ramp of the kernel-only loop below - and after 10 ms of idling the next
    # 25 launches of the headline kernel run at 1.09 instead of 1.04 ms, tools/clock_gap_probe.py. First use now:)
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
    st.synchronize()

    def step():
        if r1 <= r0:
            return          # a rank without rows (more ranks than row units) only joins the collectives
        rc = ea.lib().eu_hip_render(C.byref(tgt), srcs, nsrcs, C.c_void_p(out.data_ptr()),
                                    tw * nch * 4, 1, C.c_void_p(st.cuda_stream))
        if rc:
            raise SystemExit("render failed: " + ea.lib().eu_hip_last_error().decode())

    def sync_all():
        ea.lib().eu_hip_sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # kernel-only time of the same launches with HIP events on the library's stream, BEFORE the steps: it
    # builds the plan and brings the clocks up (after the set-up's idle gaps the first ~10 launches of a
    # process run slower: 1.32-1.36 ms per step for 1 step, 1.21 from 10 on); the timed steps
    # below carry events of their own
    kernel_ms_pre = ea.render_timed(args, sources, out.data_ptr(), max(a.steps, 20), nch, r0, r1, band) if r1 > r0 else 0.0
    # ... and on until ~0.2 s of launches have gone by: 25 launches of the 1-ms headline kernel are not enough for the
    # power management to reach the clocks it then holds (measured on one box, same build: 1.070 ms per step with
    # K = 25, 1.036 with K = 200; the kernel-only loop 1.049 over launches 21-40, 1.034 over 200). Set-up work,
    # outside the W warm-up steps and the K timed steps, which are what the contract says they are.
    if r1 > r0 and kernel_ms_pre > 0.0 and kernel_ms_pre * max(a.steps, 20) < 200.0:
        more = int(min(2000, 200.0 / kernel_ms_pre))
        if more > 0:
            kernel_ms_pre = ea.render_timed(args, sources, out.data_ptr(), more, nch, r0, r1, band)

    bar_t = torch.zeros(1, device=dev, dtype=torch.float32) if dist is not None and not rehearse else None
    trace = os.environ.get("EU_BENCH_TRACE")
    tt = [time.perf_counter()]
    for _ in range(a.warmup):
        step()
    tt.append(time.perf_counter())
    if bar_t is not None:
        with torch.cuda.stream(st):
            dist.all_reduce(bar_t)          # the closing barrier's collective, once outside the timed region
    sync_all()
    tt.append(time.perf_counter())
    ea.lib().eu_hip_launch_count.restype = C.c_ulonglong
    launches0 = ea.lib().eu_hip_launch_count()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(st)
    tt.append(time.perf_counter())
    for _ in range(a.steps):
        step()
    tt.append(time.perf_counter())
    ev1.record(st)
    if trace:
        print("bench trace (ms): warm-up enqueued %.3f, synced %.3f, first event recorded %.3f, steps enqueued %.3f" %
              tuple((tt[i + 1] - tt[i]) * 1e3 for i in range(4)), file=sys.stderr)
    # the closing barrier: over RCCL it is a one-element all-reduce queued on the launch stream BEHIND the
    # steps - it completes on a rank when every rank's stream has got there - and one synchronize waits for
    # it; a dist.barrier() after a synchronize costs two more host round trips (0.24 ms measured, against
    # 0.16 ms of kernel time per step on an eighth of the headline frame)
    if dist is not None and not rehearse:
        with torch.cuda.stream(st):
            dist.all_reduce(bar_t)
    ea.lib().eu_hip_sync()
    torch.cuda.synchronize()
    if dist is not None and rehearse:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # GPU time of exactly the timed steps, per step, on the stream they were launched on
    kernel_ms = ev0.elapsed_time(ev1) / a.steps if r1 > r0 else 0.0
    launches_per_step = (ea.lib().eu_hip_launch_count() - launches0) / a.steps if nsrcs == 1 else 1
    if dist is not None:
        tmax = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- final gather of the strips on rank 0 (outside the timed steps) -------
    gather_ms = None
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        gsrc = out.cpu() if rehearse else out
        if ranges is not None:
            frame = gather_ranges(dist, gsrc, ranges, th, tw, nch, rank, world, dst=0)
        else:
            frame = gather_bands(dist, gsrc, th, tw, nch, rank, world, BAND_ROWS, dst=0)
        torch.cuda.synchronize()
        dist.barrier()
        gather_ms = 1e3 * (time.perf_counter() - tg)
        if rehearse and rank == 0:
            whole = torch.empty((th, tw, nch), device=dev, dtype=torch.float32)
            tw_ = args.target(nch, 0, th, 0)
            rc = ea.lib().eu_hip_render(C.byref(tw_), srcs, nsrcs, C.c_void_p(whole.data_ptr()),
                                        tw * nch * 4, 1, None)
            ea.lib().eu_hip_sync()
            same = rc == 0 and torch.equal(whole.cpu().view(torch.int32), frame.view(torch.int32))
            print(f"REHEARSAL world {world}: gathered frame identical to a single launch: {same}", file=sys.stderr)
            del whole
        del frame


    # ---- the boundary with HOST buffers (never `value`): pixels in, pixels out ----
    # load = H2D of the source + prefilter/brace on the device; render_to_host =
    # kernel + D2H of the frame into pageable memory (second call: pages touched)
    host_ms = None
    pcie = None
    if world == 1 and nsrcs == 1 and not a.no_cpu_baseline:
        hbuf = np.empty((r1 - r0, tw, nch), np.float32)
        ea.render(args, src, nch, r0, r1, out=hbuf)
        th0 = time.perf_counter()
        ea.render(args, src, nch, r0, r1, out=hbuf)
        host_ms = 1e3 * (time.perf_counter() - th0)
        # what the link gives for the same bytes: a plain device -> PINNED host copy, and the
        # same into the pageable buffer the call above was given
        try:
            pin = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
            pin.copy_(out); torch.cuda.synchronize()
            tp0 = time.perf_counter(); pin.copy_(out); torch.cuda.synchronize()
            t_pin = time.perf_counter() - tp0
            hb = torch.from_numpy(hbuf)
            hb.copy_(out); torch.cuda.synchronize()
            tp0 = time.perf_counter(); hb.copy_(out); torch.cuda.synchronize()
            t_page = time.perf_counter() - tp0
            nbytes = out.numel() * 4
            pcie = {"frame_bytes": nbytes, "pinned_copy_ms": round(1e3 * t_pin, 2),
                    "pinned_GBps": round(nbytes / t_pin / 1e9, 1),
                    "pageable_copy_ms": round(1e3 * t_page, 2)}
            del pin, hb
        except Exception as e:      # no pinned memory on this box: report the render only
            pcie = {"error": str(e)[:80]}
        del hbuf

    probe = None
    if a.probe_stages:
        probe = {}
        for st in (1, 2):
            tgs = args.target(nch, r0, r1, st)
            ms = C.c_float()
            ea.lib().eu_hip_render_timed(C.byref(tgs), srcs, 1, C.c_void_p(out.data_ptr()),
                                         tw * 3 * 4, 5, C.byref(ms))
            probe[f"stage{st}_ms"] = round(ms.value, 4)
    npix_total = tw * th
    npix_rank = tw * (r1 - r0)
    ms_per_step = 1e3 * elapsed / a.steps
    value = npix_total * a.steps / elapsed / 1e6

    # algorithmic bytes of one launch on this rank: every core coefficient the
    # frame needs read once (the whole source: each rank's tile of a cubemap
    # spans all longitudes) + every output pixel of the tile written once
    # (B_alg = 4*NCH*(N_src + N_out))
    if sprj in (ea.CUBEMAP, ea.BIATAN6):
        sec = ea.cubemap_metrics(sw)["section_px"]
        n_src = sec * 6 * sec
    else:
        n_src = sw * sh
    n_src *= nsrcs
    # per rank: the WHOLE source (a rank's rows of a cubemap or spherical target span all
    # longitudes, band-interleaved shares all latitudes too) + its own output rows
    alg_bytes = 4.0 * nch * (n_src + npix_rank)
    # (a rank without rows - more ranks than row units - has launched nothing: no rate to report)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None

    # HBM traffic is a RECORD, not measured in this run: PMC counters need rocprofv3 passes of
    # their own (tools/gpu_prof.sh); profiles/traffic.json holds the last ones, with their round
    traffic, traffic_source, profile_ms, profile_frac = None, None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf) and world == 1:
        try:
            rec = json.load(open(tf))
            w_ = rec.get("workloads", {}).get(a.workload)
            if w_:
                traffic = w_.get("hbm_bytes_per_step")
                traffic_source = f"profiles/traffic.json, {rec.get('round')} at commit {rec.get('commit')}: {w_.get('kernels')}"
                # the fraction the committed rocprofv3 summary certifies (profiled runs clock lower): beside the live one
                profile_ms = w_.get("kernel_ms_profile")
                if profile_ms:
                    profile_frac = round(alg_bytes / (profile_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        except Exception:
            traffic = None

    if nsrcs > 1:
        kernel_name = "eu_render_multi_kernel"
    elif sprj in (5, 6) and degree in (2, 3) and nch in (3, 4) and not twine and os.environ.get("EU_HIP_R4", "") != "0":
        kernel_name = "eu_render4s_kernel (per-wave LDS staging of the footprint) + eu_render4d_kernel (its work list)"
    elif (sprj == 0 and tprj in (ea.CUBEMAP, ea.RECTILINEAR) and degree in (2, 3) and nch in (3, 4) and not twine
          and ypr == (0, 0, 0) and world == 1 and os.environ.get("EU_HIP_R4", "") != "0"):
        kernel_name = ("eu_render5_kernel (persistent waves, per-wave LDS staging of the footprint, 16x16 tiles on rows with a "
                       "column plan) + eu_render4d_kernel (its work list: the tiles around the poles)")
    elif sprj in (0, 5, 6) and degree in (1, 2, 3):
        kernel_name = ("eu_render2_kernel (packed two-pixel; big cubic lat/lon jobs: + eu_render3_kernel on the row "
                       "runs where source rows run across, launch-level layout choice)")
    else:
        kernel_name = "eu_render_kernel"
    result = {
        "metric": "Mpix/s reprojected (16K lat/lon->cubemap, b-spline-3)",
        "value": round(value, 1),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": WORKLOAD_TEXT[a.workload], "name": a.workload,
                   "channels": nch, "spline_degree": degree, "twine": twine,
                   "rows_per_gpu": r1 - r0,
                   "tiling": "whole frame" if world == 1 else (
                       f"contiguous strips of equal estimated cost over {world} ranks" if ranges is not None
                       else f"bands of {BAND_ROWS} rows, round-robin over {world} ranks"),
                   "gather_ms_untimed": None if gather_ms is None else round(gather_ms, 3),
                   "value_incl_gather": None if gather_ms is None else round(npix_total / (ms_per_step + gather_ms) / 1e3, 1),
                   "setup_s": round(t_setup, 2),
                   "host_boundary": {"source_load_s": round(t_load, 3),
                                     "render_to_host_ms": None if host_ms is None else round(host_ms, 1),
                                     "d2h_reference": pcie,
                                     "note": "render_to_host = kernel + D2H of the frame into the caller's pageable buffer; not part of value"}},
        "roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                     "bytes_counted": "this rank: the whole source + its own output rows" if world > 1 else "the job's algorithmic bytes",
                     "frac_of_job_bytes_over_n": None if (achieved is None or world == 1) else round(
                         4.0 * nch * (n_src + npix_total) / world / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms_profile": profile_ms, "frac_profile": profile_frac,
                     "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4),
                     "kernel_ms_note": "HIP events on the launch stream around the timed steps, per step; kernel_ms_pre: the kernel-only "
                                       "loop before the W warm-up steps (>= 0.2 s of launches: brings the clocks up)",
                     "kernel_ms_pre": round(kernel_ms_pre, 4),
                     "launches_per_step": launches_per_step,
                     "algorithmic_bytes": alg_bytes},
    }

    # ---- CPU baseline: the oracle (a port) on a band of rows, rank 0, N=1 ----
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import euo
        import jobs
        osrc = []
        for k, sk in enumerate(sources):
            cont = sk.download()
            g, _ = sk.info()
            extra = {}
            if a.workload == "config5":
                extra = dict(yaw=views[k][0], pitch=views[k][1], roll=views[k][2],
                             lens=dict(a=0.01, b=-0.03, c=0.02))
            osrc.append(jobs.oracle_source_from_container(
                sprj, sw, sh, shfov, cont, g, degree, nch,
                ea.cubemap_metrics(sw) if sprj in (ea.CUBEMAP, ea.BIATAN6) else None, **extra))
        if nsrcs == 1:
            osrc = osrc[0]
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        cores = min(cores, 64)      # a 1-GPU box's fair share of a large host
        # probe, then size the band for about --cpu-seconds of work
        probe_rows = max(1, min(th // 6, (4 * cores)))
        mid = th // 12          # inside the first cube face / upper part of the frame
        t = time.perf_counter()
        jobs.oracle_render(args, osrc, 0, mid, mid + probe_rows, nthreads=cores)
        dt = time.perf_counter() - t
        rows = int(max(probe_rows, min(th, probe_rows * a.cpu_seconds / max(dt, 1e-3))))
        rows = min(rows, th)
        # spread the band over the frame: take it from the middle
        b0 = max(0, (th - rows) // 2)
        t = time.perf_counter()
        ref = jobs.oracle_render(args, osrc, 0, b0, b0 + rows, nthreads=cores)
        dt = time.perf_counter() - t
        # the same rows from the GPU: the baseline run doubles as a full-size
        # parity spot check
        chk = ea.render(args, sources, nch, b0, b0 + rows)
        same = bool((chk.view(np.uint32) == ref.view(np.uint32)).all())
        result["cpu_baseline"] = {
            "value": round(rows * tw / dt / 1e6, 2), "unit": "Mpix/s", "cores": cores,
            "kind": "port",
            "sample": f"rows {b0}..{b0 + rows} of the same frame ({rows * tw / 1e6:.1f} Mpix, {dt:.1f} s), "
                      "oracle/eu_oracle.c with OpenMP; not envutil's SIMD binary",
            "gpu_rows_bit_identical": same,
        }
    if probe:
        result["probe"] = probe
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
