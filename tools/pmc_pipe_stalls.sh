#!/bin/bash
# tools/pmc_pipe_stalls.sh: where recon_lean_rows_kernel's wave cycles go (SQ counters; separate --pmc passes, kernel trace only): the headline
# launch (cold, one stream, 16 batches per launch) and a launch of 65 536 rows of 256 bases (tools/short_rows.py's shape)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; T=$R/gpurun_out/pmc_pipe_stalls; rm -rf $T; mkdir -p $T
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $T/h$i -- python3 $R/bench.py --no-cpu-baseline --no-hot --no-verify --streams 1 --steps 32 --warmup 16 --min-region-ms 1 --sustained-s 0 --no-secondary > $T/h$i.log 2>&1 || echo "headline group $i failed: $(tail -2 $T/h$i.log)"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $T/s$i -- python3 $R/tools/short_rows.py 65536 256 > $T/s$i.log 2>&1 || echo "short group $i failed: $(tail -2 $T/s$i.log)"
done
python3 - $T <<'PY'
import csv, sys, glob, collections
T = sys.argv[1]
for tag, name in (("h", "headline: 16 x 8192 rows of 2048 bases, one-hot, cold"), ("s", "65 536 rows of 256 bases, one-hot (fixed length)")):
    print("==", name)
    for f in sorted(glob.glob(f"{T}/{tag}*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "recon_lean_rows_kernel<true, false, false" in k or "recon_lean_rows_kernel<1, 0, 0" in k:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            v = sorted(v)
            print(f"   {c:24s} n={len(v):4d} median={v[len(v)//2]:16.1f}")
PY
rm -rf $T/h*/ $T/s*/
