#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
for a in "256 65536 150" "256 65536 900" "256 65536 1600" "256 16384 200"; do python tools/stamps_pipe.py $a; done > $O/stamps_pipe2.txt 2>&1; cat $O/stamps_pipe2.txt
