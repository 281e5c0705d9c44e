#!/bin/bash
# tools/ablate.sh: where a batch's time goes -- the default bench with parts switched off (timing only, wrong output):
# GVL_DBG 2 = no stores, 4 = no reference loads, 6 = neither; 262144 = rows with indels take the SNP-only phases A/B
# (no re-alignment, no allele bytes), 786432 = ... and no scan plan either; 65536 = re-read runs instead of re-aligning
# in LDS (correct output); 16384 = the all-purpose kernel (round 2's).  Cold (hg38 scale, rotating) and hot (64 Mbp, one batch).
# The ablations live in the diagnostic build only (the shipped library masks those bits off): built here, loaded through GVL_HIP_LIB.
cd ${GRAFT_REPO_ROOT:-/root/repo}
[ -f tools/libgvl_hip_diag.so ] || bash tools/build_diag.sh
export GVL_HIP_LIB=$PWD/tools/libgvl_hip_diag.so
for sc in "--scale hg38" "--scale hg38 --rotate 64" "--scale small --rotate 1"; do for dbg in 0 2 4 6 262144 786432 65536 16384; do echo -n "$sc GVL_DBG=$dbg: "; GVL_DBG=$dbg timeout 300 python bench.py --no-cpu-baseline --no-hot --sustained-s 0 --min-region-ms 300 --steps 200 $sc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('in flight us/step %.2f   alone %.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"; done; done
