#!/usr/bin/env python3
"""profiles/r04/traffic_table.md: per kernel of a profiled shape — launches per pass, the memory-side counter (FETCH_SIZE as
reported, WRITE_SIZE), the read factor calibrated on the kernel's access shape, calibrated and upper-bound bytes, and the
kernel's scratch / spill figures from the built library (VERDICT r3 "next round" item 2).

    python tools/traffic_table.py KEY [KEY ...] > profiles/r04/traffic_table.md      (KEY = 16bit_stereo_125000 ...)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    res = {re.sub(r"\(.*", "", k["kernel"]): k for k in json.load(open(os.path.join(ROOT, "profiles", "r04", "kernel_resources.json")))["table"]}
    print("# HBM-side traffic per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) with scratch and spill figures\n")
    print("Reads are priced at FETCH_SIZE x the factor calibrated on the kernel's dominant read shape (tools/fetch_calibrate.hip,")
    print("profiles/r04/fetch_calibration.txt): 2.0 coalesced streams and the 16-bit predictor staging (one aligned line per row and tile), 1.707 the 24-bit staging, 1.43 the entropy decoder's word stream;")
    print("`upper` is the blanket 2 x FETCH_SIZE + WRITE_SIZE of rounds 1-3.  Scratch = `.private_segment_fixed_size` per lane, spills =")
    print("`.vgpr_spill_count` / `.sgpr_spill_count` (tools/kernel_table.py); where a kernel touches scratch: tools/scratch_sites.py")
    print("(k_search1_lane<16>, k_search2_lane<16, 2>: per-pass set-up only, no scratch access inside a tile loop).")
    print("Collected after the predictor staging stopped re-reading a tile's history from the PCM (DESIGN.md section 4.0); the tables before that")
    print("change are in git history (commit 3874e34: 125 000-packet encode step 11.18 GB calibrated / 13.95 GB upper).\n")
    for key in sys.argv[1:]:
        t = traffic[key]
        per = t.get("_per_kernel", {})
        print(f"## {key} (kernel sources {traffic['_fingerprint'][key]}), bytes per encode pass\n")
        print("| kernel | launches | FETCH_SIZE (raw) | WRITE_SIZE | read factor | calibrated | upper | scratch B/lane | VGPR spills | SGPR spills |")
        print("|---|---|---|---|---|---|---|---|---|---|")
        tot = [0, 0]
        for k, v in sorted(per.items(), key=lambda kv: -kv[1]["bytes_per_pass"]):
            r = res.get(k, {})
            print(f"| `{k}` | {v['launches_per_pass']} | {v['fetch_counter_bytes'] / 1e6:.1f} MB | {v['write_bytes'] / 1e6:.1f} MB | {v['read_factor']} | "
                  f"{v['bytes_per_pass'] / 1e6:.1f} MB | {v['upper_bound_bytes'] / 1e6:.1f} MB | {r.get('private_segment_fixed_size', '-')} | "
                  f"{r.get('vgpr_spill_count', '-')} | {r.get('sgpr_spill_count', '-')} |")
            tot[0] += v["bytes_per_pass"]
            tot[1] += v["upper_bound_bytes"]
        print(f"| **step** | | | | | **{tot[0] / 1e9:.2f} GB** | **{tot[1] / 1e9:.2f} GB** | | | |\n")
        dk = "decode_" + key
        if dk in traffic:
            d = traffic[dk]
            up = d.get("_upper_bound", {})
            print(f"### decode pass of the same stream\n")
            print("| kernel | calibrated | upper |")
            print("|---|---|---|")
            s0 = s1 = 0
            for k, v in sorted(((k, v) for k, v in d.items() if isinstance(v, (int, float))), key=lambda kv: -kv[1]):
                print(f"| `{k}` | {v / 1e6:.1f} MB | {up.get(k, 0) / 1e6:.1f} MB |")
                s0 += v
                s1 += up.get(k, 0)
            print(f"| **pass** | **{s0 / 1e9:.2f} GB** | **{s1 / 1e9:.2f} GB** |\n")


if __name__ == "__main__":
    main()
