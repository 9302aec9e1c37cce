#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file (which blocks hold the MFMAs, and how many
vector / transcendental / LDS / memory instructions sit beside them).
    python tools/isa_mix.py file.s <mangled-kernel-name-substring> [min_mfma]"""
import re, sys, collections
s = open(sys.argv[1]).read()
pat = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 4
names = [n for n in re.findall(r'^(_Z\S+):', s, re.M) if pat in n]
for name in names:
    i = s.index('\n' + name + ':')
    j = s.index('.end_amdhsa_kernel', i)
    body = s[i:j]
    blocks, cur = [], ['entry', []]
    blocks.append(cur)
    for ln in body.split('\n'):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
            continue
        t = ln.strip()
        if not t or t.startswith(('.', ';')) or t.endswith(':'):
            continue
        cur[1].append(t.split()[0])
    meta = s[j - 6000:j + 200]
    print(name)
    for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_group_segment_fixed_size', '.amdhsa_private_segment_fixed_size'):
        m = re.search(re.escape(k) + r'\s+(\S+)', meta)
        if m: print('   ', k, m.group(1))
    for bn, ins in blocks:
        c = collections.Counter()
        for op in ins:
            if op.startswith('v_mfma'): c['mfma'] += 1
            elif op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): c['trans'] += 1
            elif op.startswith('v_pk_'): c['vpk'] += 1
            elif op.startswith('v_cvt'): c['cvt'] += 1
            elif op.startswith('v_mov') or op.startswith('v_accvgpr'): c['mov'] += 1
            elif op.startswith('v_'): c['valu'] += 1
            elif op.startswith('ds_'): c['ds'] += 1
            elif op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): c['vmem'] += 1
            elif op.startswith('s_waitcnt'): c['wait'] += 1
            elif op.startswith('s_barrier'): c['barrier'] += 1
            elif op.startswith('s_nop'): c['nop'] += 1
            elif op.startswith('s_'): c['salu'] += 1
            else: c['other'] += 1
        if c['mfma'] >= min_mfma:
            print('   %-10s %4d instr  %s' % (bn, len(ins), dict(c)))
