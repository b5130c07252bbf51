"""Condensed view of a bench.py JSON line:  python scripts/show_bench.py <file.json> [--entries]"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("headline %.0f rays/s  %.3f ms  roofline %.4f  eager %s  rgb_only %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"] if d.get("roofline") else -1,
      d.get("eager", {}).get("ms_per_step"), d.get("rgb_only", {}).get("ms_per_step")))
if "kernels" in d:
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]:
        print("   %-28s %6.3f ms  hbm %s" % (k, v["ms_per_step"], v.get("hbm_frac")))
for key in ("schedule_weighted", "schedule_weighted_4096_no_pose"):
    s = d.get(key)
    if s:
        print(key, s["ms_per_step"], "ms", s["rays_s"], "rays/s", [(e["epochs"], e["ms_per_step"]) for e in s["epochs_and_ms"]])
b = d.get("best_yaml_step")
if b:
    for k, v in b["regimes"].items():
        print("== %-36s %7.3f ms  %9.0f rays/s  M=%d  graphs %s  eager+events %s  device %s" % (k, v["ms_per_step"], v["rays_s"], v["samples_per_step"], v["hip_graphs"],
              v["eager_ms_per_step_with_events"], v["device_ms_per_step"]))
        if "--entries" in sys.argv:
            for e, t in sorted(v["entry_points"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
                print("   %-28s %6.3f ms x%.1f  hbm %s" % (e, t["ms_per_step"], t["calls_per_step"], t.get("hbm_frac")))
for c in d.get("configs", []):
    print("%-62s %7.3f ms  enc %s" % (c["name"][:62], c["ms_per_step"], c.get("encode_frac")))
la = d.get("with_lin_assignment")
if la:
    print("lin_assign", {k: v["ms_per_step"] for k, v in la.items() if isinstance(v, dict)})
r = d.get("render")
if r:
    print("render", {k: v["ms_per_image"] for k, v in r.items() if isinstance(v, dict)})
if d.get("mfma_util"):
    print("mfma_util", d["mfma_util"]["frac"], "sustained", d.get("sustained", {}).get("ms_per_step"))
