"""Idle time between consecutive kernels of the graph-replayed train steps, from a rocprofv3 --kernel-trace csv:
   python3 scripts/step_gaps.py <kernel_trace.csv>
A step = the kernels from one adam launch's end to the next one's end (the last steps of the file are the timed, replayed ones).  Prints per
step: wall, sum of kernel durations, sum of gaps, and the largest gaps with the kernels on either side."""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void |<.*|\(.*", "", r["Kernel_Name"])[:28]) for r in rows]
ends = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
# steps end at the LAST adam launch of each step: adam launches closer than 200 us belong together
marks = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ev[ends[k + 1]][0] - ev[i][1] > 200000]
for a, b in list(zip(marks[:-1], marks[1:]))[-6:]:
    seg = ev[a + 1:b + 1]
    wall = (seg[-1][1] - ev[a][1]) / 1e3
    busy = sum(e[1] - e[0] for e in seg) / 1e3
    gaps = [((seg[i][0] - (seg[i - 1][1] if i else ev[a][1])) / 1e3, seg[i - 1][2] if i else "adam_kernel", seg[i][2]) for i in range(len(seg))]
    tot = sum(max(g[0], 0.0) for g in gaps)
    top = sorted(gaps, key=lambda g: -g[0])[:5]
    print("step: wall %.0f us, %d kernels busy %.0f us, gaps %.0f us; largest: %s" % (wall, len(seg), busy, tot, "; ".join("%.1f (%s -> %s)" % g for g in top)))
