#!/bin/bash
# tools/collect_profiles.sh <gpurun_out tag> <rNN>: copy what is judged from a tools/profile_round.sh pass into profiles/rNN_*
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$R/gpurun_out/${1:-r05}; P=$R/profiles; N=${2:-r05}
for f in default k20_1 k20_2 k20_3 k2000_1 k2000_2 k2000_3 many1 many1_k20 nopipe_many5 nolean hot_small warm64 rotate1024 cfg2 cfg3_haps cfg4 cfg1_cpu; do
  [ -s $T/bench_$f.json ] && tail -1 $T/bench_$f.json > $P/${N}_bench_$f.json
done
for f in cfg5_epoch cfg5_epoch_hg38 cfg5_epoch_hg38_nopipe ragged_epoch_hg38 ragged_epoch_hg38_sizing_per_batch ragged_epoch_hg38_nopipe deferred_rows kbench; do
  [ -s $T/$f.txt ] && grep -v "amdgpu.ids" $T/$f.txt > $P/${N}_$f.txt
done
cp_stats() { local f=$(find $T/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${N}_$2.csv; }
cp_stats stats_default kernel_stats_default_3streams_x16
cp_stats stats_1stream kernel_stats_1stream_cold
cp_stats stats_1stream_hot kernel_stats_1stream_hot
[ -s $R/gpurun_out/${1:-r05}_profile_round.log ] && grep -v "amdgpu.ids" $R/gpurun_out/${1:-r05}_profile_round.log | tail -120 > $P/${N}_profile_round_summary.txt
ls $P | grep "^${N}_" | tr '\n' ' '
