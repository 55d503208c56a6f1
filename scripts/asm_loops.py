"""Instruction mix of every loop (backward branch) in a gfx950 assembly listing: scripts/asm_loops.py file.s"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
for i, l in enumerate(lines):
    m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if not m: continue
    t = m.group(1) or m.group(2)
    if t not in labels or labels[t] >= i: continue
    c = collections.Counter()
    for b in lines[labels[t]:i]:
        b = b.strip()
        if not b or b.startswith(('.', ';')): continue
        op = b.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if 'dpp' in b: c['dpp'] += 1
            if op.startswith(('v_fma', 'v_fmac', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_pk_')): c['fp'] += 1
            if op.startswith(('v_mov', 'v_accvgpr')): c['mov'] += 1
            if op.startswith('v_cndmask'): c['cnd'] += 1
            if op.startswith('v_pk_'): c['pk'] += 1
            if op.startswith(('v_readlane', 'v_readfirst', 'v_writelane')): c['lane'] += 1
            if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
    if sum(c.values()) > 40: print(t, labels[t], i, dict(c))
