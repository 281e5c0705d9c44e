#!/bin/bash
# tools/build_commit.sh <name> <commit>: the library of an earlier commit -> tools/lib_<name>.so (a same-box A/B against the tree's:
# GVL_HIP_LIB=tools/lib_<name>.so; tools/ab_headline.sh).  Built in a scratch worktree under /tmp.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; C=$2
W=/tmp/gvl_wt_$N
rm -rf $W; git -C $R worktree prune; git -C $R worktree add -f --detach $W $C > /dev/null 2>&1
(cd $W && python3 -c "import __graft_entry__ as g; g.build_hip()")
cp $W/genvarloader_amd/libgvl_hip.so $R/tools/lib_$N.so
git -C $R worktree remove --force $W
echo tools/lib_$N.so
