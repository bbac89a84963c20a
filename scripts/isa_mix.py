#!/usr/bin/env python3
"""Static instruction mix per kernel from hipcc -S output (developer tool).
usage: isa_mix.py file.s [kernel-substring]   -- with a substring, also prints the per-basic-block mix of that kernel"""
import re, sys, collections

def main():
    txt = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else None
    cur, blocks, kernels = None, None, {}
    for line in txt:
        m = re.match(r'(_ZN3nnn\w+):\s', line)
        if m:
            cur = m.group(1); kernels[cur] = collections.OrderedDict(); blk = 'entry'; kernels[cur][blk] = collections.Counter(); continue
        if cur is None: continue
        if line.startswith('.Lfunc_end'):
            cur = None; continue
        m = re.match(r'(\.LBB\w+):', line)
        if m:
            blk = m.group(1); kernels[cur][blk] = collections.Counter(); continue
        m = re.match(r'\s+([a-z_0-9]+)\s', line)
        if m and not m.group(1).startswith('.'): kernels[cur][blk][m.group(1)] += 1
    def summ(c):
        tot = sum(c.values())
        f = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        return f"total {tot:6d} valu {f('v_'):6d} pk {f('v_pk_'):5d} ds {f('ds_'):5d} vmem {f('global_')+f('buffer_')+f('scratch_'):5d} salu {f('s_'):5d} waitcnt {c['s_waitcnt']:4d} mfma {f('v_mfma'):4d}"
    for k, bl in kernels.items():
        c = collections.Counter()
        for b in bl.values(): c.update(b)
        if sum(c.values()) < 40: continue
        short = re.sub(r'ENS.*|EPK.*', '', k)[7:]
        print(f"{short:20s} {summ(c)}")
        if want and want in k:
            for name, b in bl.items():
                if sum(b.values()) >= 30: print(f"    {name:14s} {summ(b)}  top {b.most_common(8)}")
main()
