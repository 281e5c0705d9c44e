#!/bin/bash
# tools/resusage.sh [unit ...]: per-kernel VGPR / SGPR spills / scratch / LDS / occupancy of the current source (no GPU needed).
# unit = a translation unit of csrc/ without .hip (default: all five)
R=${GRAFT_REPO_ROOT:-/root/repo}
UNITS=${@:-gvl_hip gvl_recon gvl_lean gvl_lean_pipe gvl_tracks}
for u in $UNITS; do
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -Wno-unused-value -Wno-pass-failed -I$R/include $R/genvarloader_amd/csrc/$u.hip -o /tmp/_res_$u.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
def dem(n):
    try: return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::','')
    except Exception: return n
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r'remark:\s+(VGPRs|SGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)',l)
    if m and cur:
        d[m.group(1).replace(' [bytes/lane]','').replace(' [bytes/block]','').replace(' [waves/SIMD]','')]=m.group(2)
        if m.group(1).startswith('LDS'): print(dem(cur)[:90].ljust(90), d)
" &
done
wait
