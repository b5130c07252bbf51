#!/usr/bin/env python3
"""Which kernels lose waves per SIMD to their SGPR count?
usage: scripts/sgpr_occupancy.py file.s [file.s ...]      (hipcc -S --cuda-device-only listings)
On gfx950 a kernel whose .amdhsa_next_free_sgpr exceeds 74 (75 + VCC/flat-scratch/XNACK = 81 -> a 96-register allocation) does not get
the 8th wave on a SIMD although the runtime's occupancy calculator says it does (measured: scripts/exp/lds_occupancy.hip,
profiles/README.md round 4) - for a 1024-thread workgroup that is one workgroup per CU instead of two."""
import re
import shutil
import subprocess
import sys


def demangle(name):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return name
    return subprocess.run([tool, name], capture_output=True, text=True).stdout.strip()


rows = []
for path in sys.argv[1:]:
    txt = open(path).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        get = lambda k: int(re.search(r"\.amdhsa_%s (\d+)" % k, body).group(1))
        vgpr, sgpr, lds, acc = get("next_free_vgpr"), get("next_free_sgpr"), get("group_segment_fixed_size"), get("accum_offset")
        ma = re.search(re.escape(name) + r"\.num_agpr, (\d+)", txt)
        agpr = int(ma.group(1)) if ma else 0
        alloc = ((acc + agpr if agpr else vgpr) + 7) // 8 * 8
        by_vgpr = min(8, 512 // alloc)
        by_sgpr = 8 if sgpr <= 74 else 7 if sgpr <= 90 else 6
        rows.append((path, name, vgpr, agpr, sgpr, lds, by_vgpr, by_sgpr))
print("%d kernels; those whose SGPR count costs waves:" % len(rows))
for r in sorted(rows):
    if r[7] < r[6]:
        print("  vgpr %3d agpr %3d sgpr %3d static lds %6d  waves/SIMD by vgpr %d, by sgpr %d  %s" % (r[2], r[3], r[4], r[5], r[6], r[7], demangle(r[1])[:110]))
