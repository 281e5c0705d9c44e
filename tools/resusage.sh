#!/bin/bash
# per-kernel VGPR / scratch / LDS of the current source (no GPU needed)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -I$R/include $R/genvarloader_amd/csrc/gvl_hip.hip -o /tmp/_res.so -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)',l)
    if m and cur:
        d[m.group(1).split()[0]]=m.group(2)
        if m.group(1).startswith('LDS'): print(cur[:70].ljust(70), d)
"
