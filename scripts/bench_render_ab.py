"""render_pruned / render lines of bench.py alone: python scripts/bench_render_ab.py [--dense]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dense = "--dense" in sys.argv
args = bench.parse([a for a in sys.argv[1:] if a != "--dense"])
dev = torch.device("cuda:0")
all_ch = {"rgb", "depth", "semantics", "inst_embedding"}
out = bench.render_image_line(args, dev, all_ch, 2, pruned=not dense)
for k, v in out.items():
    if isinstance(v, dict):
        print("%-34s %8.2f ms per image  device %8.2f  %s" % (k, v["ms_per_image"], v["device_ms_per_image"], json.dumps(v["entry_points_ms_per_image"])))
