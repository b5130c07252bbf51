"""Per-launch HBM traffic per kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KB (x1024) and FETCH_SIZE
under-reports by 2x.   python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"][:110]] += float(r["Counter_Value"])
            n[r["Kernel_Name"][:110]] += 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"FETCH_SIZE_KB_per_launch": round(f, 1), "WRITE_SIZE_KB_per_launch": round(w, 1),
                  "hbm_bytes_per_launch_raw": int((f + w) * 1024), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
    import os
    out["_commit"] = os.environ.get("PAG_COMMIT")      # the commit the passes were taken at (the GPU box has no .git: passed in by the caller)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print("wrote", sys.argv[3], len(out), "kernels")


if __name__ == "__main__":
    main()
