#!/usr/bin/env python3
"""Where does a kernel touch its scratch (spill) memory?  Disassembles ONE kernel of the built library and reports, for every
loop (a backward branch target .. the branch), how many scratch_load / scratch_store instructions and how many instructions
in all it contains — i.e. whether spills sit on a per-step / per-tile path or outside the hot loops.

    python tools/scratch_sites.py 'k_search1_lane<16>'  [--so alac_amd/libalac_hip.so]
"""
import argparse
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel")
    ap.add_argument("--so", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alac_amd", "libalac_hip.so"))
    ap.add_argument("--dump", help="write the kernel's disassembly here")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="ss_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(a.so, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        for co in sorted(glob.glob(local + ".*gfx950")):
            syms = subprocess.run([f"{LLVM}/llvm-objdump", "-t", co], capture_output=True, text=True).stdout
            names = [l.split()[-1] for l in syms.splitlines() if " F .text" in l]
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
            hit = [n for n, d in zip(names, dem) if a.kernel in d.replace("alacdev::", "").replace("void ", "") and not n.endswith(".kd")]
            if not hit:
                continue
            sym = hit[0]
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={sym}", co], capture_output=True, text=True).stdout
            if a.dump:
                open(a.dump, "w").write(dis)
            ins = []  # (addr, text)
            for l in dis.splitlines():
                m = re.match(r"\s+(\S.*?)\s+//\s+([0-9A-Fa-f]+):", l)
                if m:
                    ins.append((int(m.group(2), 16), m.group(1)))
            addr_index = {ad: i for i, (ad, _) in enumerate(ins)}
            loops = []
            for i, (ad, t) in enumerate(ins):
                m = re.match(r"s_c?branch\w*\s+.*?(-?\d+)\s*$", t) or re.match(r"s_cbranch\w*\s+(\S+)", t)
                mm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", dis.splitlines()[0]) if False else None
            # objdump prints branch targets as offsets in the comment: "s_cbranch_scc1 65221 // ...: BF85FEC5 <sym+0x1234>"
            for l in dis.splitlines():
                m = re.match(r"\s+(s_c?branch\w*)\s+\S+\s+//\s+([0-9A-Fa-f]+):\s+\S+\s+<[^+>]+\+0x([0-9A-Fa-f]+)>", l)
                if m:
                    src = int(m.group(2), 16)
                    base = ins[0][0]
                    dst = base + int(m.group(3), 16)
                    if dst <= src:
                        loops.append((dst, src))
            total_sl = sum(1 for _, t in ins if t.startswith("scratch_load"))
            total_ss = sum(1 for _, t in ins if t.startswith("scratch_store"))
            print(f"{dem[names.index(sym)]}: {len(ins)} instructions, {total_sl} scratch loads, {total_ss} scratch stores")
            loops = sorted(set(loops), key=lambda x: (x[1] - x[0]))
            for dst, src in loops:
                body = [t for ad, t in ins if dst <= ad <= src]
                sl = sum(1 for t in body if t.startswith("scratch_load"))
                ss = sum(1 for t in body if t.startswith("scratch_store"))
                print(f"  loop {dst - ins[0][0]:#8x} .. {src - ins[0][0]:#8x}: {len(body):6d} instructions, {sl:4d} scratch loads, {ss:4d} scratch stores")
            return 0
        print("kernel not found", file=sys.stderr)
        return 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
