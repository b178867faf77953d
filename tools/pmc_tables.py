#!/usr/bin/env python3
"""Turn the rocprofv3 counter CSVs of tools/profile_round.sh into the tables bench.py reads:

    python tools/pmc_tables.py DIR KEY --passes N [--decode-passes M] [--fingerprint F]

  DIR/pmc_fetch_size.csv, DIR/pmc_write_size.csv   (separate --pmc FETCH_SIZE / WRITE_SIZE passes)
      -> profiles/hbm_traffic.json[KEY]            HBM bytes per encode PASS and stage: the sum over EVERY launch of the
                                                   stage's kernels in the run / N passes (a stage that launches twice per
                                                   pass, e.g. the two packet classes of the final pass, counts both)
      -> profiles/hbm_traffic.json["decode_" KEY]  the same for the decode kernels / M decode passes
  DIR/pmc_insts_0.csv (SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS), DIR/pmc_insts_1.csv (SQ_INSTS_VMEM_RD
  SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES)
      -> profiles/instruction_mix.json[KEY]        wave-instructions per pass: total, per stage, per kernel

Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: counter unit is KiB; on gfx950
FETCH_SIZE tallies a 128-B request as 64 B, so a WIDE COALESCED streaming read is 2 * FETCH_SIZE * 1024 and "other access
widths are uncalibrated: calibrate on a known byte count in your own access pattern".  tools/fetch_calibrate.hip does that
for this library's read shapes (profiles/r04/fetch_calibration.json, known bytes / counter bytes):
    stream16      64 lanes x 16 B consecutive                                  2.000   (packer, k_dec_stage, un-mix, planes)
    rows128       predictor staging, 16-bit stereo: 8 lanes cover the 128 new bytes of one packet row per tile (one aligned
                  line), rows 16 KB apart, nothing re-read                      2.000
    rows192<0>    the same for 20- / 24-bit stereo: 12 lanes, 192 bytes         1.707
                  (until the staging kept a tile's history in LDS it re-read 64 of every 192 bytes one tile later: 1.466 on
                  the calibration kernel, whose re-read follows at once and hits the L2 — in the real kernels a tile's
                  3 000+ instructions lay between and the re-read went to the fabric again)
    lane_rows16   one lane = one row, ONE 16-B load per iteration                0.444   (the counter is exact here: 64-B
                  requests, and each line is fetched 2.25 times before the lane has used it up) — no kernel of the library
                  reads like this any more; the decoder's one-lane predictors issue a line's eight loads back to back
so reads = FACTOR[kernel's dominant shape] * FETCH_SIZE * 1024 with factor 2.0 / 1.707 (20- / 24-bit staging) / 1.43 (the entropy
decoder's word stream, from its known distinct bytes) (read_factor below); every
table also carries the blanket 2 x figure rounds 1-3 reported (`_upper_bound`) and the uncorrected counter (`_counter_raw`).
WRITE_SIZE is exact for 16-B stores (calibrated 1.000); scattered 4-byte stores are counted at 64 B per request (0.113).
Every table is stored with the fingerprint of the kernel sources it was collected on (alac_amd.source_fingerprint(),
computed ON THE GPU BOX by profile_round.sh): bench.py ignores a table whose fingerprint is not its own.
"""
import argparse
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel-name substring -> bench.py stage name
STAGE_OF = [
    ("k_search1_fused", "lms_search1"), ("k_search1_lane", "lms_search1"), ("k_lms_search1", "lms_search1"),
    ("k_gol_count1", "golomb_count1"),
    ("k_search2_fused", "lms_search2"), ("k_search2_lane", "lms_search2"), ("k_lms_search2", "lms_search2"),
    ("k_gol_count2", "golomb_count2"), ("k_decide_fast", "lms_final"),
    ("k_init_state", "lms_search1"), ("k_decide1", "lms_search2"), ("k_decide2", "lms_final"),
    ("k_final_fused", "lms_final"), ("k_lms_final", "lms_final"), ("k_gol_final", "golomb_final"),
    ("k_class_count", "lms_final"), ("k_class_assign", "lms_final"), ("k_class_pred", "lms_final"),
    ("k_class_final", "lms_final"), ("k_class_coder", "golomb_final"), ("k_splice_split", "golomb_final"),
    ("k_chain_", "lms_final"),
    ("k_finalize", "finalize_scan"), ("k_scan_sizes", "finalize_scan"), ("k_pack", "pack"),
]


# dominant READ shape of a kernel -> factor on FETCH_SIZE (module docstring); default 2.0 (coalesced)
STAGING = ("k_search1_lane", "k_search2_lane", "k_class_final", "k_final_fused", "k_search1_fused", "k_lms_search", "k_lms_final")
READ_FACTOR = [
    # decoder: checked against KNOWN distinct bytes at 125 000 packets — k_dec_entropy_wide reads the 853 MB stream once
    # (counter 598 MB: 1.43); k_dec_unpc_wide reads 3.6 GB of residual rows once (counter 1.84 GB: 1.95, i.e. whole-line
    # requests: the default 2.0 applies — the one-lane-per-row shape of the calibration, one 16-byte load per iteration, is NOT
    # what this kernel does: its eight loads of a line are issued back to back)
    ("k_dec_entropy_wide", 1.43),
]


def read_factor(k):
    if any(sub in k for sub in STAGING):
        m = re.search(r"<(\d+)", k)
        return 1.707 if m and m.group(1) in ("20", "24") else 2.0  # rows192<0> / rows128 (and 32-bit: two whole lines)
    for sub, f in READ_FACTOR:
        if sub in k:
            return f
    return 2.0


def short_name(k):
    k = k.replace("void ", "").replace("alacdev::", "")
    k = re.sub(r"\(.*", "", k)
    return k.replace("(anonymous namespace)::", "")


def sums(path):
    """{kernel: {counter: sum over all dispatches}}, {kernel: dispatches}"""
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short_name(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            n[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in n.items()}


def stage_of(k):
    for sub, st in STAGE_OF:
        if sub in k:
            return st
    return None


def update(path, key, table, fp):
    try:
        with open(path) as f:
            doc = json.load(f)
    except Exception:
        doc = {}
    doc[key] = table
    doc.setdefault("_fingerprint", {})[key] = fp
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("key")
    ap.add_argument("--passes", type=int, required=True, help="encode passes in each profiled run (warmup + steps x repeats)")
    ap.add_argument("--decode-passes", type=int, default=0)
    ap.add_argument("--fingerprint", default=None)
    a = ap.parse_args()
    fp = a.fingerprint
    if fp is None:
        with open(os.path.join(a.dir, "fingerprint.txt")) as f:
            fp = f.read().strip()
    note = {"_note": "bytes (resp. wave-instructions) per PASS = sum over every launch in the profiled run / passes; reads = "
                     "READ_FACTOR(kernel) * FETCH_SIZE KiB * 1024 with the factor calibrated on the kernel's dominant read shape "
                     "(tools/fetch_calibrate.hip: 2.0 coalesced and the 16-bit predictor staging, 1.707 the 20- / 24-bit staging; 1.43 the entropy decoder's word stream), writes = "
                     "WRITE_SIZE KiB * 1024; _upper_bound = the blanket 2 x FETCH_SIZE + WRITE_SIZE of rounds 1-3; tools/pmc_tables.py"}
    fpath, wpath = os.path.join(a.dir, "pmc_fetch_size.csv"), os.path.join(a.dir, "pmc_write_size.csv")
    if os.path.exists(fpath) and os.path.exists(wpath):
        rd, _ = sums(fpath)
        wr, nl = sums(wpath)
        enc, dec, raw = defaultdict(float), defaultdict(float), {}
        encU, decU = defaultdict(float), defaultdict(float)
        for k in sorted(set(rd) | set(wr)):
            F, W = rd[k].get("FETCH_SIZE", 0.0) * 1024.0, wr[k].get("WRITE_SIZE", 0.0) * 1024.0
            f = read_factor(k)
            b, bu = f * F + W, 2.0 * F + W
            st = stage_of(k)
            if st:
                enc[st] += b / a.passes
                encU[st] += bu / a.passes
                raw[k] = {"bytes_per_pass": int(b / a.passes), "launches_per_pass": round(nl.get(k, 0) / a.passes, 2),
                          "fetch_counter_bytes": int(F / a.passes), "write_bytes": int(W / a.passes), "read_factor": f,
                          "upper_bound_bytes": int(bu / a.passes)}
            elif "k_dec" in k and a.decode_passes:
                dec[k.split("<")[0]] += b / a.decode_passes
                decU[k.split("<")[0]] += bu / a.decode_passes
        t = {k: int(v) for k, v in enc.items()}
        t["_per_kernel"] = raw
        t["_upper_bound"] = {k: int(v) for k, v in encU.items()}
        t.update(note)
        update(os.path.join(ROOT, "profiles", "hbm_traffic.json"), a.key, t, fp)
        print("traffic", a.key, {k: v for k, v in t.items() if not k.startswith("_")}, "upper", t["_upper_bound"])
        if dec:
            # the plane clears / memsets of the decode pass (fillBufferAligned) cannot be told from the encoder's small flag
            # clears by name; the encoder's are a few KB per pass, so they are all charged to the decode pass
            for k in rd:
                if "fillBuffer" in k and a.decode_passes:
                    v = (2.0 * rd[k].get("FETCH_SIZE", 0.0) + wr[k].get("WRITE_SIZE", 0.0)) * 1024.0 / a.decode_passes
                    dec["fillBufferAligned"] += v
                    decU["fillBufferAligned"] += v
            d = {k: int(v) for k, v in dec.items()}
            d["_upper_bound"] = {k: int(v) for k, v in decU.items()}
            d.update(note)
            update(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "decode_" + a.key, d, fp)
            print("traffic decode_" + a.key, {k: v for k, v in d.items() if not k.startswith("_")})
    ipaths = [os.path.join(a.dir, f"pmc_insts_{i}.csv") for i in range(2)]
    if all(os.path.exists(p) for p in ipaths):
        per = defaultdict(lambda: defaultdict(float))
        launches = {}
        for p in ipaths:
            s, nl = sums(p)
            launches.update(nl)
            for k, c in s.items():
                for name, v in c.items():
                    per[k][name] += v
        total, stages, kern = 0.0, defaultdict(float), {}
        CNT = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")
        for k, c in per.items():
            st = stage_of(k)
            if not st:
                continue
            w = sum(c.get(n, 0.0) for n in CNT) / a.passes
            total += w
            stages[st] += w
            waves = c.get("SQ_WAVES", 0.0) / a.passes
            kern[k] = {"launches_per_pass": round(launches.get(k, 0) / a.passes, 2), "waves_per_pass": int(waves),
                       "wave_instructions_per_pass": int(w),
                       "per_wave": {n: int(c.get(n, 0.0) / max(c.get("SQ_WAVES", 1.0), 1.0)) for n in CNT + ("SQ_WAVE_CYCLES",)}}
        t = {"wave_instructions_per_step": int(total), "per_stage": {k: int(v) for k, v in stages.items()}, "per_kernel": kern}
        t.update(note)
        update(os.path.join(ROOT, "profiles", "instruction_mix.json"), a.key, t, fp)
        print("instructions", a.key, t["wave_instructions_per_step"], t["per_stage"])


if __name__ == "__main__":
    main()
