#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
tools/wbench.bin > $O/wbench.txt 2>&1; cat $O/wbench.txt
python bench.py > $O/bench_default2.json 2> $O/bench_default2.err; tail -c 1500 $O/bench_default2.json; tail -n 3 $O/bench_default2.err
