#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly dump (hipcc -S --cuda-device-only).

usage: isa_mix.py file.s kernel_substring [min_barriers]
For every backward branch whose body holds at least `min_barriers` s_barrier instructions, print how many
instructions of each class the body has (the body is the straight-line text between label and branch, so
nested forward branches are counted once each).  Diagnostic only."""
import re, sys, collections

def classify(op):
    if op.startswith('v_fma_f64') or op.startswith('v_mul_f64') or op.startswith('v_add_f64') or op.startswith('v_pk_'):
        return 'valu_f64'
    if op.startswith('v_') and ('dpp' in op):
        return 'valu_dpp'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'):
        return 'valu_mov'
    if op.startswith('v_cndmask') or op.startswith('v_cmp') or op.startswith('v_min') or op.startswith('v_max'):
        return 'valu_sel'
    if op.startswith('v_'):
        return 'valu_other'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_') or op.startswith('flat_'):
        return 'vmem'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_'):
        return 'salu'
    return 'other'

def main():
    path, kern = sys.argv[1], sys.argv[2]
    minbar = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kern in l and l.rstrip().split(':')[0].startswith('_Z') and ':' in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
    labels = {}
    body = lines[start:end]
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = i
    for i, l in enumerate(body):
        m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'^\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if not m or m.group(1) not in labels: continue
        j = labels[m.group(1)]
        if j >= i: continue
        seg = body[j:i]
        ops = [s.split()[0] for s in seg if s.startswith('\t') and not s.strip().startswith(('.', ';')) and s.split()]
        nb = sum(1 for o in ops if o.startswith('s_barrier'))
        if nb < minbar: continue
        c = collections.Counter()
        for s in seg:
            if not s.startswith('\t') or s.strip().startswith(('.', ';')): continue
            t = s.split()
            op = t[0]
            k = classify(op if 'dpp' not in s else op + '_dpp')
            c[k] += 1
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith('valu'))
        print(f'loop {m.group(1)} lines {start+j}-{start+i}: {tot} instr, {valu} VALU, barriers {nb}')
        print('   ', dict(sorted(c.items())))
        oc = collections.Counter(s.split()[0] for s in seg if s.startswith('\tv_'))
        print('    top VALU ops:', oc.most_common(14))

main()
