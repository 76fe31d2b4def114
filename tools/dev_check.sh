#!/bin/bash
# Development loop without a GPU: host emulation build + its Deflate tests, the library, the ISA of kernels.hip with the
# register / LDS / scratch figures of one kernel.  Usage: dev_check.sh [kernel-name-substring] [pytest files...]
R=/root/repo; K=${1:-swc_inflate_sync_kernel}; shift
T=${@:-tests/test_lane_emulation_deflate.py}
cd $R/tests/host_emu && g++ -O2 -g -std=c++17 -DSWC_HOST_EMULATION -fPIC -shared -Wno-unknown-pragmas -pthread -o libswc_emu.so emu.cpp 2>&1 | head -30
cd $R && timeout 900 python -m pytest $T -x -q 2>&1 | tail -3
cd $R && python -m swcompression_amd.build 2>&1 | grep -E "error|libswc" | tail -3
mkdir -p /tmp/asm && cd $R/swcompression_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only kernels.hip -o /tmp/asm/k.s 2>&1 | grep -E "error" | head
cd $R && python - "$K" <<'PY'
import re, sys
s = open('/tmp/asm/k.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    if sys.argv[1] in m.group(1):
        body = m.group(2)
        g = lambda k: re.search(r'\.amdhsa_%s (\S+)' % k, body).group(1)
        print(m.group(1)[:60], 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
PY
