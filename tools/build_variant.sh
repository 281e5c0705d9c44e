#!/bin/bash
# tools/build_variant.sh <name> <unit> [hipcc flags / -D...]: an A/B build tools/lib_<name>.so = the tree's objects (build/hip, after
# __graft_entry__.build()) with ONE translation unit recompiled with extra flags (or from SRC_OVERRIDE=<dir> holding replacement .inc/.hip files).
# Load it with GVL_HIP_LIB=tools/lib_<name>.so (tools/ab_cfg4.sh, tools/ab_track_libs.sh).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; U=$2; shift 2
W=$(mktemp -d)
cp $R/genvarloader_amd/csrc/* $W/
[ -n "$SRC_OVERRIDE" ] && cp $SRC_OVERRIDE/* $W/
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -I$R/include "$@" -c $W/$U.hip -o $W/$U.o
OBJS=""
for u in gvl_hip gvl_recon gvl_lean gvl_lean_pipe gvl_tracks gvl_svar2; do
  if [ $u = $U ]; then OBJS="$OBJS $W/$U.o"; else OBJS="$OBJS $R/build/hip/$u.hip.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/tools/lib_$N.so
rm -rf $W
echo "tools/lib_$N.so"
