#!/bin/bash
# the rocprofv3 / PMC half of tools/profile_round.sh alone (the bench lines come from a full pass): tools/profile_stats_only.sh <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${1:-prof}
mkdir -p $T
[ -x $R/tools/kbench.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $R/tools/kbench.hip -o $R/tools/kbench.bin
sed -n '/^cd \/tmp$/,$p' $R/tools/profile_round.sh > /tmp/_stats_half.sh
T=$T R=$R bash /tmp/_stats_half.sh
