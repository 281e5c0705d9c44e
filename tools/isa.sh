#!/bin/bash
# tools/isa.sh <unit> [kernel-name-substring]: device assembly of one translation unit of csrc/ -> /tmp/<unit>.s; prints one kernel's body size + resources
R=${GRAFT_REPO_ROOT:-/root/repo}
U=${1:-gvl_lean}
/opt/rocm/bin/hipcc $ISA_DEFS -O3 --offload-arch=gfx950 -std=c++17 -I$R/include -S --cuda-device-only $R/genvarloader_amd/csrc/$U.hip -o /tmp/$U.s -Wno-unused-value -Wno-pass-failed 2>&1 | grep -v "warning: argument unused" | head
K=${2:-recon_lean_kernel}
awk -v k="$K" '$0 ~ "^_Z.*"k".*:" {p=1} p {print} p && /^\.Lfunc_end/ {exit}' /tmp/$U.s > /tmp/kernel.s
echo "instructions: $(grep -cE '^\s+[sv]_|^\s+(ds|global|flat|buffer)_' /tmp/kernel.s)  (VALU $(grep -cE '^\s+v_' /tmp/kernel.s), SALU $(grep -cE '^\s+s_' /tmp/kernel.s), waitcnt $(grep -c s_waitcnt /tmp/kernel.s))"
grep -A40 "amdhsa_kernel .*$K" /tmp/$U.s | grep -E "next_free_vgpr|next_free_sgpr|group_segment|private_segment_fixed"
grep -E "sgpr_spill_count|vgpr_spill_count|\.name:.*$K" /tmp/$U.s | grep -A2 "$K" | head -6
