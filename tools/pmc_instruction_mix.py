#!/usr/bin/env python3
"""Merge the counter CSVs of tools/pmc_instruction_mix.sh into per-wave means per kernel (JSON on stdout)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

d = sys.argv[1]
out = {}
for mode in ("fused", "unfused"):
    acc = defaultdict(lambda: defaultdict(list))
    for path in sorted(glob.glob(os.path.join(d, f"{mode}_*.csv"))):
        per_dispatch = defaultdict(dict)
        with open(path) as f:
            for row in csv.DictReader(f):
                per_dispatch[(row["Dispatch_Id"], row["Kernel_Name"])][row["Counter_Name"]] = float(row["Counter_Value"])
        for (_, k), c in per_dispatch.items():
            for name, v in c.items():
                acc[k][name].append(v)
    res = {}
    for k, c in acc.items():
        if "alacdev" not in k:
            continue
        short = re.sub(r"\(.*", "", k.replace("void ", "").replace("alacdev::", ""))
        m = {n: sum(v) / len(v) for n, v in c.items()}
        waves = m.get("SQ_WAVES", 0)
        if not waves:
            continue
        e = {"waves": round(waves)}
        for n, v in sorted(m.items()):
            if n != "SQ_WAVES":
                e[n] = round(v / waves)
        res[short] = e
    out[mode] = res
out["_note"] = ("rocprofv3 --pmc (three passes), per-wave means (counter / SQ_WAVES) over the launches of "
                "python3 bench.py --steps 2 --warmup 1 --cpu-packets 0; SQ_WAVE_CYCLES, SQ_BUSY_CYCLES and SQ_WAIT_* are in "
                "units of 4 clock cycles; 10 000 16-bit stereo packets, encode then decode")
print(json.dumps(out, indent=1))
