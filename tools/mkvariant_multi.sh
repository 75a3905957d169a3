#!/bin/bash
# a build-time variant of the NCH=4 multi-facet kernels linked into a library of its own (A/B runs with EU_HIP_LIB):
#   tools/mkvariant_multi.sh NAME "-DFLAG=..."   ->  envutil_amd/build/libeu_hip_NAME.so ; ISA in /tmp/isa/var_NAME/
set -e
cd "$(dirname "$0")/../envutil_amd"
NAME=$1; FLAGS=$2
T=/tmp/isa/var_$NAME; mkdir -p $T
/opt/rocm/bin/hipcc $FLAGS -DEU_MULTI_NCH=4 -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-pass-failed -save-temps=obj -c csrc/eu_render_multi.hip -o $T/eu_render_multi_4.o
OBJS=$(ls build/eu_*.o | grep -v "build/eu_render_multi_4.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libeu_hip_$NAME.so $OBJS $T/eu_render_multi_4.o
python3 - $T <<'PY'
import re, sys
s=open(sys.argv[1]+'/eu_render_multi-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    n=m.group(1); b=m.group(2)
    if 'Li4ELi1ELb1ELb0ELb0ELb0' in n:
        print(sys.argv[1], re.findall(r'\.(private_segment_fixed_size|sgpr_spill_count|vgpr_count|vgpr_spill_count):\s+(\d+)', b))
PY
