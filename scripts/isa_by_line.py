#!/usr/bin/env python3
"""Instruction counts of one kernel by basic block and source-line range, from device assembly built with -g (.loc directives):
which loop of the source a run of instructions belongs to, and how many vector / LDS / scalar instructions it holds -- multiply by the
loop's trip count (known from the source) for the dynamic count.  usage: isa_by_line.py kg.s kernel_symbol_substring [src_file_substring]"""
import collections
import re
import sys

path, kname = sys.argv[1], sys.argv[2]
srcsub = sys.argv[3] if len(sys.argv) > 3 else "nnn_kernels.hip"
lines = open(path).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kname, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks = []   # (label, [(op, line)])
cur = ("entry", [])
loc = None
for l in lines[start + 1:end + 1]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
        continue
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        f = files.get(int(m.group(1)), "")
        loc = int(m.group(2)) if srcsub in f else -int(m.group(2))
        continue
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if m and not l.strip().startswith(".") and not l.strip().startswith(";"):
        cur[1].append((m.group(1), loc, l))
blocks.append(cur)
label_idx = {b[0]: i for i, b in enumerate(blocks)}
# loops: a backward branch from block j to label at index i <= j
loops = []
for j, (lab, ins) in enumerate(blocks):
    for op, loc, raw in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            t = raw.split()[-1]
            if t in label_idx and label_idx[t] <= j:
                loops.append((label_idx[t], j))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")): return "wait"
    return "salu"
def summarize(i0, i1):
    c = collections.Counter()
    ls = []
    for lab, ins in blocks[i0:i1 + 1]:
        for op, loc, raw in ins:
            c[cls(op)] += 1
            if loc and loc > 0: ls.append(loc)
    return c, (min(ls), max(ls)) if ls else (0, 0)
print(f"{len(blocks)} basic blocks; loops (innermost listed too): block range, source lines, instruction classes")
for (i0, i1) in sorted(set(loops)):
    c, (l0, l1) = summarize(i0, i1)
    print(f"  loop blocks {i0:4d}..{i1:4d}  src {l0}-{l1}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
c, _ = summarize(0, len(blocks) - 1)
print("whole kernel:", dict(c))
# per source line histogram (valu only), top lines
h = collections.Counter()
for lab, ins in blocks:
    for op, loc, raw in ins:
        if cls(op) == "valu" and loc: h[loc] += 1
print("VALU by source line (top 40):", sorted(h.items(), key=lambda kv: -kv[1])[:40])
