"""Phase between the two streams of the pipelined schedule, from a rocprofv3 kernel trace:
    python tools/phase.py <kernel_trace.csv>
For every launch of the update kernel that opens a step (sgd_update_pipe*), prints the time since the previous
one (the other stream's step start) as a fraction of the two-step period, and what ran concurrently."""
import csv
import re
import sys
from collections import Counter

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
ups = [(s, q) for s, e, n, q in rows if n.startswith("sgd_update_pipe")]
ups = ups[len(ups) // 2: len(ups) // 2 + 400]          # steady state
d = [(b[0] - a[0]) / 1e3 for a, b in zip(ups, ups[1:])]
pair = [d[i] + d[i + 1] for i in range(0, len(d) - 1, 2)]
period = sum(pair) / len(pair)
frac = [d[i] / (d[i] + d[i + 1]) for i in range(0, len(d) - 1, 2)]
print("step starts: %d; two-step period %.1f us; offset of the second stream: mean %.2f of the period "
      "(min %.2f, max %.2f)" % (len(ups), period, sum(frac) / len(frac), min(frac), max(frac)))
# which kernels overlap with which (time-weighted)
t0, t1 = ups[0][0], ups[-1][0]
act = [(s, e, n) for s, e, n, q in rows if e > t0 and s < t1]
ov = Counter()
for i, (s, e, n) in enumerate(act):
    for s2, e2, n2 in act[i + 1:]:
        if s2 >= e:
            break
        ov[tuple(sorted((n, n2)))] += (min(e, e2) - s2) / 1e3
tot = sum(ov.values())
for k, v in ov.most_common(14):
    print("  %5.1f %%  %s  |  %s" % (100 * v / tot, k[0], k[1]))

# ---- per queue: where the idle time sits (gap in front of each kernel, mean over the steady-state window)
from collections import defaultdict
byq = defaultdict(list)
for s, e, n, q in rows:
    if t0 <= s < t1:
        byq[q].append((s, e, n))
for q, ks in byq.items():
    if len(ks) < 100:
        continue
    gaps, durs = defaultdict(list), defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        gaps[n1].append(max(0, s1 - e0) / 1e3)
        durs[n1].append((e1 - s1) / 1e3)
    tot_g = sum(sum(v) for v in gaps.values())
    tot_d = sum(sum(v) for v in durs.values())
    nst = sum(1 for _, _, n in ks if n.startswith("sgd_update_pipe"))
    print("queue %s: %d steps, kernels %.1f us/step, idle %.1f us/step" % (q, nst, tot_d / nst, tot_g / nst))
    for n in sorted(gaps, key=lambda k: -sum(gaps[k])):
        print("    gap before %-32s %6.1f us (kernel %6.1f us)" % (n, sum(gaps[n]) / len(gaps[n]), sum(durs[n]) / len(durs[n])))
