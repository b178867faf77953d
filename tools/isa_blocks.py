#!/usr/bin/env python3
"""Basic-block instruction histogram of one kernel in a hipcc -S dump: tools/isa_blocks.py FILE.s KERNEL_SUBSTR [top]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E", ":")) or (l.startswith("_Z") and key in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur)); cur = []; name = l.split(":")[0]
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
print("kernel lines", start, end, "instructions", sum(len(b) for _, b in blocks))
for name, b in sorted(blocks, key=lambda x: -len(x[1]))[:top]:
    h = collections.Counter(b)
    cls = collections.Counter()
    for m, c in h.items():
        k = "valu" if m.startswith("v_") else "salu" if m.startswith("s_") else "lds" if m.startswith("ds_") else "vmem" if m.startswith(("global_", "buffer_", "flat_")) else "other"
        if m.startswith("s_waitcnt"): k = "waitcnt"
        if m.startswith("s_nop"): k = "nop"
        cls[k] += c
    print(f"\n{name}: {len(b)} instr  {dict(cls)}")
    print("  " + ", ".join(f"{m}:{c}" for m, c in h.most_common(40)))

if len(sys.argv) > 4 and sys.argv[4] == "flow":
    print("\n--- blocks in order (name, instrs, valu, last instr) ---")
    body = lines[start + 1:end]
    name, cnt, valu, last = "entry", 0, 0, ""
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        t = l.strip()
        if m:
            print(f"{name:12s} {cnt:5d} {valu:5d}  {last}")
            name, cnt, valu, last = m.group(1), 0, 0, ""
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cnt += 1
        valu += t.startswith("v_")
        if t.startswith(("s_cbranch", "s_branch")):
            last += t.replace("\t", " ") + " | "
    print(f"{name:12s} {cnt:5d} {valu:5d}  {last}")
