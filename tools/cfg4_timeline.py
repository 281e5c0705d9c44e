"""tools/cfg4_timeline.py <kernel_trace.csv> [t0_us] [span_us]: the GPU timeline of a stretch of the cfg4 step from a rocprofv3 --kernel-trace csv:
per kernel its queue, start (us since the stretch began), duration; then busy / idle time of the stretch (union of kernel intervals)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
# the stretch: the LAST third of the trace by default (steady state), `span` us long
span = float(sys.argv[3]) if len(sys.argv) > 3 else 1500.0
T0 = ks[0][0]
t_begin = (float(sys.argv[2]) * 1e3 + T0) if len(sys.argv) > 2 and float(sys.argv[2]) >= 0 else ks[len(ks) * 2 // 3][0]
sel = [k for k in ks if k[0] >= t_begin and k[0] < t_begin + span * 1e3]
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n[:n.index("(")][:48] if "(" in n else n[:48]
for s, e, n, q in sel:
    print(f"{(s - t_begin) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q:>3s}  {short(n)}")
iv = sorted((s, e) for s, e, _, _ in sel)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = iv[-1][1] - iv[0][0] if len(iv) > 1 else 1
print(f"stretch {tot / 1e3:.1f} us: busy {busy / 1e3:.1f} us, idle {(tot - busy) / 1e3:.1f} us in {len(gaps)} gaps; the largest: "
      + ", ".join(f"{(b - a) / 1e3:.1f} us at {(a - t_begin) / 1e3:.0f}" for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:6]))
n_step = sum(1 for k in sel if "realign_paint_kernel" in k[2])
print(f"{n_step} track kernels in the stretch = {tot / 1e3 / max(1, n_step):.1f} us per batch")
