#!/usr/bin/env python3
"""Where k_qp4's scratch accesses sit: per loop (backward branch) and per barrier-separated segment.  Diagnostic only.
usage: spills4.py [asm file]   (default: compiles mpcmp.hip to /tmp/m.s)"""
import re, subprocess, sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = sys.argv[1] if len(sys.argv) > 1 else '/tmp/m.s'
if len(sys.argv) <= 1:
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-Xclang', '-target-feature', '-Xclang', '-load-store-opt',
                           '-DMPCMP_SPLIT_N25', *os.environ.get('EXTRA', '').split(), '-S', '--cuda-device-only', '-o', asm, os.path.join(ROOT, 'mpc_motion_planner_amd/csrc/mpcmp.hip')],
                          stderr=subprocess.DEVNULL)
s = open(asm).read()
name = os.environ.get('KNAME', '_ZN5mpcmp5k_qp4ILi4EEEv12mpcmp_configNS_2WSEPKjiPKd')
i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
lines = s[i:j].split('\n')
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
print(len(lines), 'lines; scratch ops total', sum(1 for l in lines if l.strip().startswith('scratch_')))
seen = set()
for n, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
    if not m: continue
    t = m.group(1) or m.group(2)
    if t in labels and labels[t] < n and (t not in seen):
        seg = lines[labels[t]:n]
        nb = sum(1 for x in seg if 's_barrier' in x)
        if nb < int(os.environ.get('MINB', '7')): continue
        seen.add(t)
        ns = sum(1 for x in seg if x.strip().startswith('scratch_'))
        print('loop', t, 'lines', labels[t], n, 'len', n - labels[t], 'barriers', nb, 'scratch', ns)
        k = 0; cnt = collections.Counter(); ins = collections.Counter()
        for x in seg:
            x = x.strip()
            if 's_barrier' in x: k += 1
            if x.startswith('scratch_'): cnt[k] += 1
            if x and not x.startswith(('.', ';')) and not x.endswith(':'): ins[k] += 1
        print('   instructions per segment:', dict(ins))
        print('   scratch per segment     :', dict(cnt))
