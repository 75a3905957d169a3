#!/usr/bin/env python3
"""profiles/r03_headline_isa.txt: what the shipped headline kernel is made of. Compiles eu_render4.hip (which
includes eu_render5.h) with -save-temps, takes eu_render5_kernel<3,3,SPHERICAL,FAST> out of the ISA listing and
reports the code-object metadata, the instruction mix (whole kernel and per stage of the 16x16 tile, cut at the
stage's first characteristic instruction), and - from the PMC record in profiles/ - the executed counts per tile.
usage: python tools/isa_report.py > profiles/r03_headline_isa.txt"""
import collections, os, re, subprocess, sys, tempfile, json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_Z17eu_render5_kernelILi3ELi3ELi0ELb1EEv"
tmp = tempfile.mkdtemp(prefix="isa_")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
                       "-Wno-unused-function", "-Wno-pass-failed", "-save-temps=obj", "-c",
                       os.path.join(ROOT, "envutil_amd/csrc/eu_render4.hip"), "-o", os.path.join(tmp, "eu_render4.o")],
                      stderr=subprocess.DEVNULL)
lines = open(os.path.join(tmp, "eu_render4-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
s = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
e = next(i for i in range(s, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = lines[s:e]
ins = [(i, l.split()[0]) for i, l in enumerate(body) if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(".")]
meta = [l.strip() for l in body if re.search(r"amdhsa_(next_free_vgpr|next_free_sgpr|group_segment_fixed_size|private_segment_fixed_size|accum_offset)", l)]
occ = [l.strip() for l in lines[s:e + 200] if "; Occupancy" in l or "; NumVgprs" in l or "; NumSgprs" in l or "; ScratchSize" in l or "; LDSByteSize" in l][:6]

def cls(op):
    if op.startswith("v_pk_"): return "VALU packed fp32 (v_pk_*)"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt")): return "VALU transcendental"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "VALU lane<->scalar (incl. SGPR spills)"
    if "_dpp" in op: return "VALU DPP"
    if op.startswith("v_cvt") or "f64" in op: return "VALU f64 / conversions"
    if op.startswith("v_mov"): return "VALU moves"
    if op.startswith("v_"): return "VALU other"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_load_lds",)): return "LDS-DMA"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")): return "vector memory"
    if op.startswith("s_load"): return "scalar memory"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")): return "branches"
    return "SALU other"

print(f"# eu_render5_kernel<3,3,SPHERICAL,FAST> at commit {subprocess.check_output(['git','-C',ROOT,'rev-parse','--short','HEAD']).decode().strip()} (hipcc -O3 --offload-arch=gfx950 -ffp-contract=off)")
print("## code-object metadata")
for m in meta + occ: print("  ", m)
print("   waves per SIMD by registers: 512 / 128 = 4 (launch bounds 256 x 4); workgroups per CU by LDS: 163840 / 39936 = 4 -> 16 waves per CU")
tot = collections.Counter(cls(op) for _, op in ins)
print(f"## static instruction mix of the whole kernel ({len(ins)} instructions: two loops, two copies of each tile body)")
for k, v in tot.most_common(): print(f"   {v:6d}  {k}")
# the first copy of the 16x16 tile: from the loop head's column-table use to its last store
first_dpp = next(i for i, (_, op) in enumerate(ins) if "_dpp" in op)
stores = [i for i, (_, op) in enumerate(ins) if op.startswith("global_store")]
dmas = [i for i, (_, op) in enumerate(ins) if op.startswith("global_load_lds")]
dsr = [i for i, (_, op) in enumerate(ins) if op.startswith("ds_read_b128")]
end16 = next(i for i in stores if i > first_dpp and sum(1 for j in stores if first_dpp < j <= i) == 4)
tile_dma = [i for i in dmas if i > first_dpp][:2]
taps0 = next(i for i in dsr if i > tile_dma[-1])
# head of the tile body: walk back from the DPP block to the previous store / loop label
prev_store = max([i for i in stores if i < first_dpp] + [0])
stages = [("coordinates (table values -> ray y, two latitude chains, md_to_spline, gate, split)", prev_store, first_dpp),
          ("box (DPP reduction, lane reads, scalar fit / gate tests)", first_dpp, tile_dma[0]),
          ("LDS-DMA issue + y weights of both pairs + window addresses", tile_dma[0], taps0),
          ("taps of both pairs (32 + 32 ds_read_b128, 2 x 140 packed operations), next tile's table loads, stores", taps0, end16 + 1)]
print("## first copy of the 16x16 tile body (256 pixels per wave), instructions between stage markers [static; the rare branches")
print("##   (work-list exit, unaligned tails of the DMA loop) are inside the ranges, the DMA loop body counts once]")
tsum = collections.Counter()
for name, a, b in stages:
    c = collections.Counter(cls(op) for _, op in ins[a:b])
    valu = sum(v for k, v in c.items() if k.startswith("VALU"))
    tsum.update(c)
    print(f"   {name}\n      VALU {valu} (packed {c['VALU packed fp32 (v_pk_*)']}, moves {c['VALU moves']}, lane<->scalar {c['VALU lane<->scalar (incl. SGPR spills)']}, DPP {c['VALU DPP']}, f64/cvt {c['VALU f64 / conversions']}, trans {c['VALU transcendental']}), "
          f"SALU {c['SALU other']}, LDS {c['LDS']}, LDS-DMA {c['LDS-DMA']}, vmem {c['vector memory']}, waitcnt {c['s_waitcnt']}, nop {c['s_nop']}, branches {c['branches']}")
valu_tile = sum(v for k, v in tsum.items() if k.startswith("VALU"))
print(f"   total VALU of the tile body: {valu_tile} per 256 pixels = {valu_tile / 2:.0f} per 128 pixels")
# longest dependent chain: the latitude chain (atan2f with x > 0) between the table values and the base position
print("## longest dependent chain per pixel (operations that each need the previous one's result; eu_math2.h):")
print("   ray y 2 | range test 3 | division y/x: rcp + 7 = 8 | table index 3 + LDS round trip | num/den 3 | division 8 | z, w 2 |")
print("   polynomial s1: 11 (s2's 9 run beside it) | s1 + s2, x * (..) 2 | (xs - lo) - x, hi - .. 3 | sign 1 | f64 subtract 3 |")
print("   FMA division by the extent 5 | * total, - .5, - offset 3 | gate 2 | floor, subtract, convert 3  =  ~62 dependent operations,")
print("   ~50 of them packed (8 cycles each from one wave): ~450 cycles; a 16x16 tile runs two such chains side by side (pairs A, B)")
try:
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["workloads"]["headline"]
    print("## executed counts (profiles/r03_headline_kernel_stats_pmc.txt, per launch of eu_render5_kernel; 786 432 16x8-tile equivalents):")
    for l in open(os.path.join(ROOT, "profiles", "r03_headline_kernel_stats_pmc.txt")):
        if "eu_render5_kernel" in l and any(k in l for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "TCP_TOTAL", "SQ_LDS_BANK", "SQ_LDS_IDX")):
            m = re.search(r"(\S+)\s+n=\s*\d+ mean=(\S+)", l)
            v = float(m.group(2))
            print(f"   {m.group(1):32s} {v:14.4g}   per 128 pixels: {v / 786432:9.1f}")
except Exception as ex:
    print("## (no PMC record found:", ex, ")")
