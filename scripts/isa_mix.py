#!/usr/bin/env python3
"""Static instruction mix of the kernels, from device-only assembly of the product sources (no GPU needed).  The transforms, the
synthesis and the fused back end are straight-line code per stream-frame apart from short loops, so their static count is what a wave
issues per stream-frame.  usage: scripts/isa_mix.py [--json out.json] [source tree (default: this repository)] [kernel name substrings ...]
(--json: the table as {"kernels": {name: {"total", "valu", "lds", "vmem", "salu", "arith", "moves", "selects", "integer"}}}; __graft_entry__.build()
writes profiles/r6_isa_mix.json with it, which bench.py's issue ceiling reads for the straight-line kernels)"""
import collections
import os
import re
import subprocess
import json
import sys

argv = sys.argv[1:]
json_out = None
if "--json" in argv:
    i = argv.index("--json")
    json_out = argv[i + 1]
    del argv[i:i + 2]
root = argv[0] if argv and os.path.isdir(argv[0]) else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
want = [a for a in argv if not os.path.isdir(a)] or ["k_fft_xp", "k_synth", "k_pitch", "k_hp", "k_rnn_wf", "k_back"]
src = os.path.join(root, "nnnoiseless_amd", "csrc")
out = "/tmp/nnn_isa_mix.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", out,
                       "-x", "hip", os.path.join(src, "nnn_batch.hip"), "-I", src, '-DNNN_WEIGHTS_PATH="x"', "-w"])
text = open(out).read()
GROUPS = (("valu", ("v_",)), ("lds", ("ds_",)), ("vmem", ("global_", "buffer_", "flat_", "scratch_")), ("salu", ("s_",)))
table = {}
print(f"{'kernel':44s} {'total':>6s} {'valu':>6s} {'lds':>5s} {'vmem':>5s} {'salu':>5s}   arithmetic (mul / add / fma, packed counted once) | moves | selects | integer + address")
for m in re.finditer(r"\n(_Z\w+):\s*; @", text):
    sym = m.group(1)
    name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.split("(")[0].replace("void ", "").replace("nnn::", "")
    if not any(w in name for w in want):
        continue
    body = text[m.end():text.index(".Lfunc_end", m.end())]
    ins = [l.split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
    c = collections.Counter(ins)
    g = collections.Counter()
    for k, v in c.items():
        g[next((n for n, p in GROUPS if k.startswith(p)), "other")] += v
    arith = sum(v for k, v in c.items() if re.match(r"v_(pk_)?(mul|add|sub|fma|fmac|mac|mad)_f(32|64)", k))
    # SIMD cycles the arithmetic alone takes per wave at the measured issue costs (profiles/r6_valu_issue.txt): plain two-operand f32 2.27, packed
    # or three-operand 4.2
    arith_cycles = sum(v * (4.2 if re.match(r"v_(pk_|fma|fmac|mac|mad)", k) else 2.27) for k, v in c.items() if re.match(r"v_(pk_)?(mul|add|sub|fma|fmac|mac|mad)_f(32|64)", k))
    mov = sum(v for k, v in c.items() if k.startswith(("v_mov", "v_pk_mov", "v_accvgpr")))
    sel = sum(v for k, v in c.items() if k.startswith("v_cndmask"))
    integer = sum(v for k, v in c.items() if re.match(r"v_(add|sub|lshl|lshr|ashr|and|or|xor|mul_lo|mul_u|mad_u|mad_i|bfe|lshl_add|add3|lshl_or)", k) and "_f" not in k)
    print(f"{name:44s} {len(ins):6d} {g['valu']:6d} {g['lds']:5d} {g['vmem']:5d} {g['salu']:5d}   {arith:5d} | {mov:4d} | {sel:4d} | {integer:4d}")
    table[name] = {"total": len(ins), "valu": g["valu"], "lds": g["lds"], "vmem": g["vmem"], "salu": g["salu"], "arith": arith, "arith_cycles": round(arith_cycles, 1), "moves": mov, "selects": sel, "integer": integer}
if json_out:
    json.dump({"what": "static instruction mix per kernel (scripts/isa_mix.py; gfx950, -O3 -ffp-contract=off -fno-slp-vectorize)", "kernels": table}, open(json_out, "w"), indent=1)
