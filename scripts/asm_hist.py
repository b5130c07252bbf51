#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing.
usage: asm_hist.py file.s <substring of the mangled kernel name> [--top N] [--dump out.s]
Counts are static (per occurrence in the listing); loops are listed with their label so hot bodies can be read off."""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and l.rstrip().endswith(":") or (l.startswith("_Z") and ": ;" in l)]
    sel = [i for i in starts if pat in lines[i]]
    if not sel:
        print("no kernel matching", pat)
        return 1
    for s in sel:
        e = next((i for i in range(s + 1, len(lines)) if lines[i].strip().startswith("s_endpgm")), len(lines))
        # function may have several s_endpgm; take until .Lfunc_end
        e = next((i for i in range(s + 1, len(lines)) if lines[i].startswith(".Lfunc_end")), e)
        body = lines[s:e]
        if dump:
            open(dump, "w").write("\n".join(body))
        hist = collections.Counter()
        for l in body:
            t = l.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            m = re.match(r"([a-z_0-9]+)", t)
            if m:
                hist[m.group(1)] += 1
        total = sum(hist.values())
        cls = collections.Counter()
        for k, v in hist.items():
            c = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if k.startswith("ds_") else "other"
            cls[c] += v
        print(lines[s][:150])
        print("  total %d  " % total + "  ".join("%s %d" % kv for kv in cls.most_common()))
        for k, v in hist.most_common(top):
            print("    %-28s %d" % (k, v))
    return 0


if __name__ == "__main__":
    sys.exit(main())
