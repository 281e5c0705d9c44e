#!/bin/bash
# CPU-only: the oracle (the checker itself) under AddressSanitizer + UBSan over the reference goldens and KATs.
# The instrumented library is built NEXT TO the tree's one and selected with GVL_ORACLE_LIB, so a failing
# run cannot leave an ASan build behind as oracle/libgvl_oracle.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -fPIC -shared -pthread $R/oracle/gvl_oracle*.c -o $T/libgvl_oracle_asan.so -lm
GVL_ORACLE_LIB=$T/libgvl_oracle_asan.so ASAN_OPTIONS=detect_leaks=0 \
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest $R/tests/test_oracle_golden.py $R/tests/test_oracle_kats.py $R/tests/test_oracle_tracks.py -x -q
