#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
t0=$(date +%s); python bench.py > $O/bench_default3.json 2> $O/bench_default3.err; echo "bench.py took $(( $(date +%s) - t0 )) s"; tail -c 300 $O/bench_default3.json
