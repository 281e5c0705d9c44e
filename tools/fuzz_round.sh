#!/bin/bash
# tools/fuzz_round.sh: the round's fuzz campaign on the current build -> stdout (profiles/rNN_fuzz.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${N:-20000}
SEED_OFFSET=${SEED_OFFSET:-0}      # (another campaign over other cases: SEED_OFFSET=1000 ...)
run() { echo -n "$1 GVL_DBG=${2:-0} ${3}: "; env GVL_DBG=${2:-0} $3 python $1 ${4:-$N} $(( ${5:-4001} + SEED_OFFSET )) 2>&1 | tail -1; }
echo "== fuzz_lean.py (rows of one chunk; one-hot / one-hot + bytes / bytes)"
for d in 0 65536 32768 33554432 33619968 33587200 67108864 1073741824; do run tools/fuzz_lean.py $d "" $N 4001; done
echo "== fuzz_lean.py FUZZ_LONG=1 (rows of 2 052 ... 40 000 bases: the chunked form)"
for d in 0 32768 65536 536870912; do run tools/fuzz_lean.py $d "FUZZ_LONG=1" $((N / 8)) 4002; done
run tools/fuzz_lean.py 0 "FUZZ_LONG=1 FUZZ_SUB=1" $((N / 8)) 4003; run tools/fuzz_lean.py 0 "FUZZ_LONG=1 FUZZ_SUB=4" $((N / 8)) 4004
echo "== fuzz_lean.py FUZZ_RAGGED=1 (output_length -1: the pipelined kernel's ragged form; with FUZZ_LONG=1 the chunked kernel's)"
for d in 0 33554432 67108864; do run tools/fuzz_lean.py $d "FUZZ_RAGGED=1" $N 4010; done
for d in 0 32768 65536 1048576; do run tools/fuzz_lean.py $d "FUZZ_LONG=1 FUZZ_RAGGED=1" $((N / 8)) 4011; done
run tools/fuzz_lean.py 0 "FUZZ_LONG=1 FUZZ_RAGGED=1 FUZZ_SUB=1" $((N / 8)) 4012
echo "== fuzz_lean.py FUZZ_MIXED=1 (round 6: ragged batches of short rows with a few long ones on the pipelined kernel: front workgroups / long rows at the waves' ends)"
for d in 0 256 33554432; do run tools/fuzz_lean.py $d "FUZZ_MIXED=1" $((N / 4)) 4015; done
echo "== fuzz_lean.py FUZZ_MANY=1 (4-16 batches of 1 500-6 000 queries in one grid, default flags; rows per wave 1 / 1.5 / 2 / 3 / 8; channel-major, annotated)"
run tools/fuzz_lean.py 0 "FUZZ_MANY=1" $((N / 40)) 4020; run tools/fuzz_lean.py 0 "FUZZ_MANY=1 FUZZ_RAGGED=1" $((N / 40)) 4021
run tools/fuzz_lean.py 1073741824 "FUZZ_MANY=1" $((N / 80)) 4022
echo "== fuzz.py (ragged / fixed, keep masks, annotations, both layouts: ragged + no keep + row-major now takes the pipelined RAG kernel)"
for d in 0 33554432 67108864 8 134217728; do run tools/fuzz.py $d "" $N 4005; done
echo "== fuzz_tracks.py, fuzz_fused_tracks.py"
run tools/fuzz_tracks.py 0 "" $((N / 4)) 4006; run tools/fuzz_tracks.py 8 "" $((N / 8)) 4007
run tools/fuzz_fused_tracks.py 0 "" $((N / 10)) 4008; run tools/fuzz_fused_tracks.py 2097152 "" $((N / 10)) 4009
run tools/fuzz_fused_tracks.py 268435456 "" $((N / 10)) 4013      # (no row plans: every chunk walks)
run tools/fuzz_fused_tracks.py 1073741824 "" $((N / 10)) 4014     # (the general track kernel for every chunk)
echo "== fuzz_svar2.py (round 6: the SVAR2 two-source provider -- merge + the kernels over the merged table vs the oracle's provider)"
for d in 0 64 80 8 16384 67108864; do run tools/fuzz_svar2.py $d "" $N 4030; done
