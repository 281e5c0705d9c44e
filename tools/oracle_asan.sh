#!/bin/bash
# CPU-only: the oracle (the checker itself) under AddressSanitizer + UBSan over the reference goldens and KATs.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/oracle/libgvl_oracle.so /tmp/libgvl_oracle_backup.so 2>/dev/null || true
(cd $R/oracle && gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c11 -fPIC -shared -pthread gvl_oracle*.c -o libgvl_oracle.so -lm)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest $R/tests/test_oracle_golden.py $R/tests/test_oracle_kats.py $R/tests/test_oracle_tracks.py -x -q
rm -f $R/oracle/libgvl_oracle.so
python -c "import sys; sys.path.insert(0, '$R'); from oracle import oracle; oracle.build()"
