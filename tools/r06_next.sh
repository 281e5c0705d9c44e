#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in 1536 2048 3072; do for d in 0 256 0 256; do echo -n "pairs $p GVL_DBG=$d: "; GVL_DBG=$d python tools/spliced_bench.py $p 2>&1 | grep -E "\"kernel_ms|exon rows" | tr '\n' ' ' | sed 's/"workload": "spliced haplotypes under the exonic keep mask: //' | cut -c1-200; echo; done; done
