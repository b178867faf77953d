#!/usr/bin/env python3
"""Per-kernel resource table of the gfx950 code objects embedded in a built libalac_hip.so: VGPRs, spilled VGPRs / SGPRs,
scratch (.private_segment_fixed_size), LDS.  No GPU needed (llvm-objdump --offloading + llvm-readelf --notes).

    python tools/kernel_table.py [--so alac_amd/libalac_hip.so] [--json out.json] [--scratch-only]
"""
import argparse
import collections
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so):
    tmp = tempfile.mkdtemp(prefix="kt_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so, local)  # llvm-objdump writes the bundles next to its input
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        out = []
        for co in sorted(glob.glob(local + ".*gfx950")):
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+(?:- )?\.(\w+):\s+(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                # a kernel's map is sorted by key: ... .symbol, .uses_dynamic_stack, .vgpr_count, .vgpr_spill_count,
                # .wavefront_size closes it
                if k in ("symbol", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                         "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count"):
                    cur[k] = v
                elif k == "wavefront_size":
                    out.append(cur)
                    cur = {}
        names = [k.get("symbol", "").replace(".kd", "") for k in out]
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        for k, d in zip(out, dem):
            k["kernel"] = d.replace("void ", "").replace("alacdev::", "")
            for f in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                k[f] = int(k.get(f, 0))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--so", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alac_amd", "libalac_hip.so"))
    ap.add_argument("--json")
    ap.add_argument("--scratch-only", action="store_true")
    ap.add_argument("--grep", default="")
    a = ap.parse_args()
    ks = kernels(a.so)
    fam = collections.Counter(re.sub(r"[<(].*", "", k["kernel"]) for k in ks)
    rows = [k for k in ks if (not a.scratch_only or k["private_segment_fixed_size"]) and a.grep in k["kernel"]]
    rows.sort(key=lambda k: (-k["private_segment_fixed_size"], k["kernel"]))
    print(f"{len(ks)} kernels, {sum(1 for k in ks if k['private_segment_fixed_size'])} with scratch, "
          f"{os.path.getsize(a.so) / 1e6:.1f} MB")
    print("scratch  vspill sspill vgpr  lds    kernel")
    for k in rows:
        print(f"{k['private_segment_fixed_size']:7d} {k['vgpr_spill_count']:6d} {k['sgpr_spill_count']:6d} {k['vgpr_count']:5d} "
              f"{k['group_segment_fixed_size']:6d}  {k['kernel'][:150]}")
    print("families:", ", ".join(f"{n} x{c}" for n, c in fam.most_common()))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(kernels=len(ks), so_bytes=os.path.getsize(a.so), families=dict(fam),
                           table=[{x: k[x] for x in ("kernel", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                      "private_segment_fixed_size", "group_segment_fixed_size")} for k in ks]), f, indent=0)


if __name__ == "__main__":
    sys.exit(main())
