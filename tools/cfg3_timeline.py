"""tools/cfg3_timeline.py <kernel_trace.csv> [steps per region = 20] [batches per launch = 16]: the driver's default schedule (bench.py --steps 20:
regions of 20 steps = one launch of 16 batches + one of 4, three streams, a gate kernel in front of every region) read off a
rocprofv3 --kernel-trace: per region the first launch's start to the last launch's end, / 20 = us per step -- the number bench.py's
HIP events give -- and a sample of consecutive launches with their queues, so that the in-flight figure follows from timestamps."""
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
G = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
            for r in rows)
rec = [k for k in ks if "recon_lean_rows_kernel" in k[2]]
print(f"{len(rec)} recon_lean_rows_kernel launches; grids {sorted(set(k[4] for k in rec))}")
big = max(k[4] for k in rec)
# a region of the pipelined leg = a launch of G batches (the big grid) followed by the launch of the K - G left over (a smaller grid),
# on two different queues; the single-stream legs (kernel alone, hot) use one queue only: keep regions whose two launches sit on different queues
regions = []
i = 0
while i + 1 < len(rec):
    a, b = rec[i], rec[i + 1]
    if a[4] == big and b[4] != big and a[3] != b[3]:
        regions.append((a, b))
        i += 2
    else:
        i += 1
if not regions:
    sys.exit("no (16 + 4)-batch regions on two queues found")
span = [(max(a[1], b[1]) - min(a[0], b[0])) / 1e3 for a, b in regions]
d16 = [(a[1] - a[0]) / 1e3 for a, _ in regions]
d4 = [(b[1] - b[0]) / 1e3 for _, b in regions]
lag = [(b[0] - a[0]) / 1e3 for a, b in regions]
med = st.median(span)
print(f"{len(regions)} regions of {K} steps ({G} + {K - G} batches, two queues)")
print(f"region: first launch's start -> last launch's end: median {med:.1f} us = {med / K:.2f} us per step "
      f"(p10 {sorted(span)[len(span) // 10]:.1f}, p90 {sorted(span)[len(span) * 9 // 10]:.1f})")
print(f"the {G}-batch launch alone: median {st.median(d16):.1f} us = {st.median(d16) / G:.2f} us per batch; the {K - G}-batch launch: median {st.median(d4):.1f} us "
      f"= {st.median(d4) / (K - G):.2f} us per batch; it starts {st.median(lag):.1f} us after the first (median): {st.median(d16) - st.median(lag):.1f} us of overlap")
print(f"=> per step: ({st.median(d16):.1f} + what of the second launch sticks out) / {K}; fill / drain of a {G} + {K - G} split = "
      f"{(med - st.median(d16)):.1f} us of the region")
print("a stretch of consecutive regions (us since the first one's start):")
t0 = regions[len(regions) // 2][0][0]
for a, b in regions[len(regions) // 2: len(regions) // 2 + 6]:
    for k in (a, b):
        print(f"  {(k[0] - t0) / 1e3:9.1f} -> {(k[1] - t0) / 1e3:9.1f}  +{(k[1] - k[0]) / 1e3:6.1f} us  q{k[3]:>3s}  grid {k[4]}")
