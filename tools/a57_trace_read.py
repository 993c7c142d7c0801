"""Reads a rocprofv3 kernel-trace csv: the last `n` repetitions of the periodic pattern -> per-queue busy time, union busy time, the longest kernels' placement."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# lock-step boundaries: the policy kernel runs once per lock-step
marks = [int(r["Start_Timestamp"]) for r in rows if "k_a57_policy" in r["Kernel_Name"]]
t0, t1 = marks[-n_steps - 1], marks[-1]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
per = (t1 - t0) / n_steps / 1e3
print("period %.1f us over %d lock-steps, %d kernels per lock-step" % (per, n_steps, len(sel) / n_steps))
byq = defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs if cs is not None else 0)


for q, iv in sorted(byq.items()):
    names = defaultdict(float)
    for s, e, k in iv:
        names[(re.findall(r"k_\w+", k) or [k[:30]])[0]] += (e - s) / n_steps / 1e3
    top = sorted(names.items(), key=lambda kv: -kv[1])[:6]
    print("queue %s: %d kernels / lock-step, busy %.1f us / lock-step; top: %s" % (q, len(iv) / n_steps, union([(s, e) for s, e, _ in iv]) / n_steps / 1e3,
                                                                               ", ".join("%s %.0f" % kv for kv in top)))
allv = [(s, e) for iv in byq.values() for s, e, _ in iv]
print("any queue busy %.1f us / lock-step (idle %.1f)" % (union(allv) / n_steps / 1e3, per - union(allv) / n_steps / 1e3))
if len(sys.argv) > 3:  # one lock-step's timeline
    a, b = marks[-3], marks[-2]
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if a <= s < b:
            print("%8.1f %7.1f q%s %s" % ((s - a) / 1e3, (e - s) / 1e3, r["Queue_Id"], (re.findall(r"k_\w+(?:<[^>]*>)?", r["Kernel_Name"]) or [r["Kernel_Name"][:40]])[0]))
