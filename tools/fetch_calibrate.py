#!/usr/bin/env python3
"""Summarise tools/fetch_calibrate.sh: FETCH_SIZE / WRITE_SIZE (KiB) per launch of each calibration kernel against the bytes the
kernel is known to move -> the factor to apply to the counter for that access shape."""
import csv
import json
import os
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc, n = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void ", ""))
            acc[k] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) * 1024 for k in acc}


def main():
    d = sys.argv[1]
    expect = {}
    for line in open(os.path.join(d, "expect.txt")):
        m = re.match(r"EXPECT (\S+) (read|write) (\d+)(?: requested (\d+))?", line)
        if m:
            expect[m.group(1)] = (m.group(2), int(m.group(3)), int(m.group(4)) if m.group(4) else None)
    fetch = per_kernel(os.path.join(d, "pmc_fetch.csv"), "FETCH_SIZE")
    write = per_kernel(os.path.join(d, "pmc_write.csv"), "WRITE_SIZE")
    ms = {}
    ks = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(ks):
        for row in csv.DictReader(open(ks)):
            ms[re.sub(r"\(.*", "", row["Name"].replace("void ", ""))] = float(row["AverageNs"]) / 1e6
    out = {}
    print(f"{'kernel':22s} {'known bytes':>14s} {'counter bytes':>14s} {'known/counter':>14s} {'ms':>8s} {'GB/s':>8s}")
    for k, (kind, nbytes, req) in expect.items():
        c = (fetch if kind == "read" else write).get(k)
        if c is None:
            continue
        t = ms.get(k)
        print(f"{k:22s} {nbytes:14d} {c:14.0f} {nbytes / c:14.3f} {t or 0:8.3f} {(nbytes / t / 1e6) if t else 0:8.0f}" +
              (f"   (requested incl. re-reads {req}: {req / c:.3f})" if req else ""))
        out[k] = dict(kind=kind, known_bytes=nbytes, counter_bytes=c, factor=nbytes / c, requested_bytes=req, ms=t)
    json.dump(out, open(os.path.join(d, "calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
