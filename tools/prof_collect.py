#!/usr/bin/env python3
"""Assemble profiles/r03_<workload>_kernel_stats_pmc.txt and profiles/traffic.json from the runs of
tools/prof_all.sh (gpurun_out/r03_prof_<workload>/). usage: python tools/prof_collect.py <commit>"""
import csv, glob, json, os, re, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else "?"
STEPS = {"trace": 10 + 2}          # bench.py --steps 10 --warmup 2 (+ the kernel-only loop of >= 20 launches)
out = {"note": "HBM bytes per render step from separate rocprofv3 --pmc passes (tools/gpu_prof.sh: FETCH_SIZE and WRITE_SIZE "
               "each in a pass of its own), per launch, summed over the launches of a step. gfx950 correction "
               "(MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests as 64 B: doubled; WRITE_SIZE as read. Counters "
               "are in KB. kernel_ms_profile: sum over the step's kernels of their average duration under "
               "rocprofv3 --kernel-trace --stats (profiled runs clock lower than unprofiled ones).",
       "round": "r03", "commit": commit, "workloads": {}}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r03_prof_*"))):
    wl = d.split("r03_prof_")[1]
    summ = os.path.join(d, "summary.txt")
    if not os.path.exists(summ):
        continue
    # launches per step of each render kernel: from the kernel trace of a pmc pass (steps 3 + warmup 1 + >= 20 timed launches)
    stats = {}
    for f in glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "eu_render" in r["Name"]:
                stats[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]))
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "eu_render" in r["Kernel_Name"]:
                ctr[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    stats = {k: v for k, v in stats.items() if "colplan" not in k}     # plan pre-passes: once per job, not per step
    if not stats:
        continue
    ncalls = min(c for c, _ in stats.values())
    per_step = {k: round(c / ncalls) for k, (c, _) in stats.items()}
    kernel_ms = sum(per_step[k] * stats[k][1] for k in stats) / 1e6
    fetch = sum(per_step[k] * (sum(ctr[k]["FETCH_SIZE"]) / max(1, len(ctr[k]["FETCH_SIZE"]))) for k in stats)
    write = sum(per_step[k] * (sum(ctr[k]["WRITE_SIZE"]) / max(1, len(ctr[k]["WRITE_SIZE"]))) for k in stats)
    # FETCH_SIZE's factor for this kernel's reads: 2.0 for coalesced 16-byte-per-lane streaming (the guide's case,
    # reproduced by tools/calib_fetch.hip's stream16: 1.612 GB read, 787 059 KB counted); 1.68 for the staging shape of
    # eu_render5_kernel - LDS-DMA, 16 bytes per lane at a 12-byte stride, box rows (calib_fetch.hip's dma_boxes: 1.610 GB
    # of texels read once, 935 800 KB counted)
    def factor(k):
        return 1.68 if "eu_render5" in k else 2.0
    fetch_true = sum(per_step[k] * factor(k) * (sum(ctr[k]["FETCH_SIZE"]) / max(1, len(ctr[k]["FETCH_SIZE"]))) for k in stats)
    ent = {"kernels": " + ".join(f"{per_step[k]} x {k}" for k in stats),
           "kernel_ms_profile": round(kernel_ms, 4),
           "FETCH_SIZE_KB": round(fetch), "WRITE_SIZE_KB": round(write),
           "fetch_factor": {k: factor(k) for k in stats},
           "hbm_bytes_per_step": int((fetch_true + write) * 1024) if fetch and write else None}
    out["workloads"][wl] = ent
    dst = os.path.join(ROOT, "profiles", f"r03_{wl}_kernel_stats_pmc.txt")
    with open(dst, "w") as o:
        o.write(f"# tools/gpu_prof.sh, workload {wl}, commit {commit}: rocprofv3 --kernel-trace --stats of\n"
                f"# python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload {wl}, then one --pmc pass per group\n")
        o.write(open(os.path.join(d, "progress.txt")).read() if os.path.exists(os.path.join(d, "progress.txt")) else "")
        o.write(open(summ).read())
        o.write(f"# per step: {ent['kernels']} = {ent['kernel_ms_profile']} ms under the profiler\n")
    print(wl, ent)
if not out["workloads"]:
    sys.exit("no gpurun_out/r03_prof_* runs with a summary found: profiles/ left as it is")
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
