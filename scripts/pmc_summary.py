"""Mean PMC counter value per kernel from the per-set outputs of scripts/pmc_kernel.sh:
   python scripts/pmc_summary.py <outdir> [kernel-name-substring ...]"""
import collections
import csv
import glob
import sys

res = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        res[r["Kernel_Name"][:90]].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
pats = sys.argv[2:]
for k, v in sorted(res.items()):
    if "at::native" in k or (pats and not any(p in k for p in pats)):
        continue
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-34s n=%3d mean %.4g" % (c, len(vals), sum(vals) / len(vals)))
