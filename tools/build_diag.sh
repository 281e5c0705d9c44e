#!/bin/bash
# tools/build_diag.sh: the DIAGNOSTIC build tools/libgvl_hip_diag.so = every translation unit with -DGVL_DIAG: per-phase time stamps
# (tools/stamps*.py) and the timing ablations GVL_DBG 1 / 2 / 4 / 262144 / 524288 / 8388608 / 16777216, which change the OUTPUT and
# therefore do not exist in the shipped library (csrc/gvl_internal.inc: DBG_ABLATIONS).  Load it with
# GVL_HIP_LIB=tools/libgvl_hip_diag.so (tools/ablate.sh, tools/pmc_realign_parts.sh do).  Never the library under test.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$(mktemp -d)
OBJS=""
for u in $R/genvarloader_amd/csrc/*.hip; do
  b=$(basename $u .hip)
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -I$R/include -DGVL_DIAG -c $u -o $W/$b.o &
  OBJS="$OBJS $W/$b.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/tools/libgvl_hip_diag.so
rm -rf $W
echo "tools/libgvl_hip_diag.so"
