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
