#!/usr/bin/env python3
"""Turn two rocprofv3 counter CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass) into the per-stage
HBM byte table bench.py reads (profiles/hbm_traffic.json).

    python tools/pmc_to_traffic.py FETCH.csv WRITE.csv KEY [OUT.json]

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: counter unit is
KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import csv
import json
import sys
from collections import defaultdict

# kernel-name substring -> bench.py stage name
STAGE_OF = [
    ("k_search1_fused", "lms_search1"), ("k_lms_search1", "lms_search1"), ("k_gol_count1", "golomb_count1"),
    ("k_lms_search2", "lms_search2"), ("k_gol_count2", "golomb_count2"),
    ("k_final_fused", "lms_final"), ("k_lms_final", "lms_final"), ("k_gol_final", "golomb_final"),
    ("k_class_count", "lms_final"), ("k_class_assign", "lms_final"), ("k_class_pred", "lms_final"), ("k_class_coder", "golomb_final"),
    ("k_finalize", "finalize_scan"), ("k_scan_sizes", "finalize_scan"), ("k_pack", "pack"),
]


def mean_per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch, write, key = sys.argv[1], sys.argv[2], sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/hbm_traffic.json"
    rd, wr = mean_per_kernel(fetch, "FETCH_SIZE"), mean_per_kernel(write, "WRITE_SIZE")
    stages, raw = defaultdict(float), {}
    for name in sorted(set(rd) | set(wr)):
        short = name.split("(")[0].split("::")[-1].split("<")[0]
        raw[short] = {"FETCH_SIZE": rd.get(name, 0.0), "WRITE_SIZE": wr.get(name, 0.0)}
        for sub, stage in STAGE_OF:
            if sub in name:
                stages[stage] += 2.0 * rd.get(name, 0.0) * 1024.0 + wr.get(name, 0.0) * 1024.0
                break
    try:
        with open(out) as f:
            doc = json.load(f)
    except Exception:
        doc = {}
    doc["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (python3 bench.py --steps 3 "
                    "--warmup 1 --cpu-packets 0), mean per launch, counter unit KiB; reads = 2 * FETCH_SIZE * 1024 "
                    "(gfx950 counts 128-B requests as 64 B), WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section). "
                    "Produced by tools/pmc_to_traffic.py from the CSVs under profiles/.")
    doc[key] = {k: int(v) for k, v in stages.items()}
    doc["raw_KiB_" + key] = raw
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc[key]))


if __name__ == "__main__":
    main()
