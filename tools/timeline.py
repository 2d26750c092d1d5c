"""Timeline of a rocprofv3 --kernel-trace CSV (tools/timeline.sh): overlap statistics of the two streams of the
pipelined schedule and one step pair printed kernel by kernel."""
import csv
import os
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
ev.sort()
# steady window: the last 60 % of the launches of the most frequent kernel
names = defaultdict(int)
for e in ev:
    names[e[2]] += 1
top = max(names, key=names.get)
marks = [e[0] for e in ev if e[2] == top]
# (TL_LO / TL_HI: another slice of the launches -- bench.py's legs follow each other: set-up and timed enqueue-only steps first)
lo, hi = marks[int(len(marks) * float(os.environ.get('TL_LO', 0.3)))], marks[int(len(marks) * float(os.environ.get('TL_HI', 0.9)))]
win = [e for e in ev if e[0] >= lo and e[1] <= hi]
nsteps = sum(1 for e in win if e[2] == top)
print("window: %.1f us, %d launches of %s -> %.2f us per step" % ((hi - lo) / 1e3, nsteps, top, (hi - lo) / 1e3 / nsteps))
# sweep
pts = []
for s, e, n, q in win:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort()
active = defaultdict(int)
depth_time = defaultdict(float)
combo_time = defaultdict(float)
prev = pts[0][0]
cur = []
for t, d, n in pts:
    dt = t - prev
    if dt > 0:
        depth_time[len(cur)] += dt
        combo_time[tuple(sorted(cur))] += dt
    prev = t
    if d > 0:
        cur.append(n)
    else:
        cur.remove(n)
tot = sum(depth_time.values())
for k in sorted(depth_time):
    print("  %d kernels on the GPU: %5.1f %% (%.1f us per step)" % (k, 100 * depth_time[k] / tot, depth_time[k] / 1e3 / nsteps))
print("combinations (us per step):")
for c, t in sorted(combo_time.items(), key=lambda x: -x[1])[:24]:
    print("  %6.2f  %s" % (t / 1e3 / nsteps, " + ".join(c) if c else "(idle)"))
# per kernel: mean duration in the window
dur = defaultdict(list)
for s, e, n, q in win:
    dur[n].append(e - s)
print("kernel durations in the window (us):")
for n, d in sorted(dur.items(), key=lambda x: -sum(x[1])):
    print("  %-36s n %5d  avg %7.1f  sum/step %6.1f" % (n, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e3 / nsteps))
# one step pair
mid = win[len(win) // 2][0]
print("timeline from the middle of the window (us, queue, kernel):")
for s, e, n, q in win:
    if mid <= s < mid + 400e3:
        print("  %8.1f %8.1f  q%-3s %s" % ((s - mid) / 1e3, (e - mid) / 1e3, q, n))
