#!/usr/bin/env python3
"""Idle GPU time between the launches of back-to-back passes, from a rocprofv3 --kernel-trace CSV (start / end stamps per launch):

    rocprofv3 --kernel-trace --output-format csv -d OUT -o kt -- python3 tools/ab_timing.py 125000 16 "" --decode --rounds 2 --passes 3 --no-oracle
    python tools/launch_gaps.py OUT/kt_kernel_trace.csv k_dec_header [PASS]  (ANCHOR = a kernel that runs once per pass)

Prints one pass (the middle one, or pass number PASS of the trace): per launch its start relative to the pass, its duration and the gap behind the launch before it,
then the pass period (anchor to anchor), the sum of its kernels and the difference = idle time per pass.
Round 4 found with it: 44 us per decode pass in front of and between two tiny hipMemsetAsync blits (now cleared by the pass's first
kernel), 5-12 us behind every stage event while the library's stage timing is on, nothing else."""
import csv
import sys


def main():
    path, anchor = sys.argv[1], sys.argv[2]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if len(idx) < 3:
        print("need at least three passes in the trace")
        return 1
    k = int(sys.argv[3]) if len(sys.argv) > 3 else len(idx) // 2
    a, b = idx[k], idx[k + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = int(rows[a - 1]["End_Timestamp"]) if a else t0
    busy = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void ", "").replace("alacdev::", "").split("(")[0]
        print(f"{(s - t0) / 1000:10.1f} us  {(e - s) / 1000:9.1f} us  gap {max(0, s - prev_end) / 1000:6.1f} us  {name[:70]}")
        busy += e - max(s, prev_end) if e > prev_end else 0
        prev_end = max(prev_end, e)
    period = int(rows[b]["Start_Timestamp"]) - t0
    print(f"pass period {period / 1000:.1f} us, kernels busy {busy / 1000:.1f} us, idle {(period - busy) / 1000:.1f} us")
    return 0


if __name__ == "__main__":
    sys.exit(main())
