#!/bin/bash
# Builds a VARIANT of libswc_hip.so here (hipcc cross-compiles; the GPU box's minutes are not spent compiling):
#   tools/build_variant.sh <name> "<-D flags>"  ->  swcompression_amd/variants/libswc_<name>.so
# Only kernels.hip is recompiled; the host objects of the last regular build are linked in.  Select it on the box with SWC_LIB=<path>.
set -e
R=/root/repo; N=$1; F=$2; D=$R/swcompression_amd/variants; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c $R/swcompression_amd/csrc/kernels.hip -o $D/kernels_$N.o -Wall -Wno-unused-function $F
OBJS=$(ls $R/swcompression_amd/build/*.o | grep -v kernels.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libswc_$N.so $D/kernels_$N.o $OBJS
rm -f $D/kernels_$N.o
echo $D/libswc_$N.so
# registers / scratch of the phase-1 kernel of the variant (scratch > 0 = spills)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only $R/swcompression_amd/csrc/kernels.hip -o /tmp/asm/k_$N.s $F 2>/dev/null && python - /tmp/asm/k_$N.s <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    if 'inflate_sync' in m.group(1) or 'lz_resolve' in m.group(1):
        g = lambda k: re.search(r'\.amdhsa_%s (\S+)' % k, m.group(2)).group(1)
        print('  ', m.group(1)[9:36], 'vgpr', g('next_free_vgpr'), 'scratch', g('private_segment_fixed_size'))
PY
