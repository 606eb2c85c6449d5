#!/usr/bin/env python3
"""Basic-block table of one kernel in a device assembly listing (developer tool):
    hipcc ... --cuda-device-only -S unit.hip -o unit.s ; python tools/asm_blocks.py unit.s <mangled-prefix>
instructions, scratch stores / loads (spills), transcendental, global and LDS-atomic counts per block + branches."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
def new(lbl): return {'label': lbl, 'n': 0, 'st': 0, 'ld': 0, 'exp': 0, 'sqrt': 0, 'glob': 0, 'ds_add': 0, 'ds': 0, 'br': []}
cur = new('entry'); stats = []
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        stats.append(cur); cur = new(m.group(1)); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur['n'] += 1
    cur['st'] += t.startswith('scratch_store'); cur['ld'] += t.startswith('scratch_load')
    cur['exp'] += t.startswith('v_exp') or t.startswith('v_ldexp'); cur['sqrt'] += t.startswith(('v_sqrt_f64', 'v_rsq_f64', 'v_rcp_f64'))
    cur['glob'] += t.startswith('global_'); cur['ds_add'] += t.startswith('ds_add'); cur['ds'] += t.startswith('ds_')
    m = re.match(r's_c?branch\S*\s+(\.LBB\d+_\d+)', t)
    if m: cur['br'].append(m.group(1))
stats.append(cur)
print("label         instr  sp_st sp_ld  exp rcp/sqrt glob  ds ds_add  branches")
for c in stats:
    print(f"{c['label']:12s} {c['n']:6d} {c['st']:6d} {c['ld']:5d} {c['exp']:4d} {c['sqrt']:6d} {c['glob']:5d} {c['ds']:4d} {c['ds_add']:5d}   {' '.join(c['br'])}")
print("total", sum(c['n'] for c in stats), "spill stores", sum(c['st'] for c in stats), "loads", sum(c['ld'] for c in stats))
