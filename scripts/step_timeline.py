"""Print the kernel sequence of the LAST training step of a rocprofv3 --kernel-trace csv (start-ordered, with the idle gap before
each launch).  usage: python3 scripts/step_timeline.py <kernel_trace.csv> [marker-substring] [steps-back]   (default marker: the optimiser's last launch, adam_kernel;
steps-back k: the step k before the last one - bench.py ends with K eager steps for the roofline kernel, the graph-replayed ones come before them)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# a step ends with the last optimizer launch of a burst
bursts = [i for k, i in enumerate(ends) if k + 1 == len(ends) or ends[k + 1] - i > 8]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lo, hi = bursts[-2 - back] + 1, bursts[-1 - back] + 1
prev_end = int(rows[lo - 1]["End_Timestamp"])
t0 = int(rows[lo]["Start_Timestamp"])
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("at::native::", "")
    print("%8.1f us  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:110]))
    busy += e - s
    prev_end = max(prev_end, e)
print("launches %d  span %.3f ms  busy %.3f ms" % (hi - lo, (prev_end - t0) / 1e6, busy / 1e6))
