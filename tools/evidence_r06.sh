#!/bin/bash
# tools/evidence_r06.sh: (a) a rocprofv3 --kernel-trace of the driver's default schedule (bench.py --steps 20) -> profiles-ready timeline summary;
# (b) FETCH_SIZE / WRITE_SIZE passes of the cfg4 step's kernels (tools/pmc_cfg4.sh).  Outputs under gpurun_out/r06/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --sustained-s 0 --no-hot > $O/trace_cfg3.json 2> $O/trace_cfg3.err
f=$(find $O/trace_cfg3 -name "*kernel_trace.csv" | head -1)
python3 $R/tools/cfg3_timeline.py $f > $O/cfg3_timeline.txt 2>&1
python3 -c "import json; d=json.loads(open('$O/trace_cfg3.json').read().strip().splitlines()[-1]); print('bench.py under the profiler: ms_per_step', d['ms_per_step'], 'kernel_ms per launch', d['roofline']['kernel_ms'])" >> $O/cfg3_timeline.txt
s=$(find $O/trace_cfg3 -name "*kernel_stats.csv" | head -1)
head -4 $s | cut -c1-220 > $O/cfg3_trace_kernel_stats.csv
rm -rf $O/trace_cfg3
bash $R/tools/pmc_cfg4.sh r06/pmc_cfg4 > $O/pmc_cfg4.txt 2>&1
rm -rf $O/pmc_cfg4/pmc_cfg4_FETCH_SIZE $O/pmc_cfg4/pmc_cfg4_WRITE_SIZE
cat $O/cfg3_timeline.txt $O/pmc_cfg4.txt
