"""Per-launch durations of the kernels whose name contains a pattern, in launch order, from a rocprofv3 --kernel-trace csv:
   python3 scripts/kernel_durations.py <kernel_trace.csv> <pattern> [<pattern> ...]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for pat in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    print(pat, len(d), "launches:", " ".join("%.0f" % x for x in d))
