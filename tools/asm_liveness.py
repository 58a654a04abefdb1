#!/usr/bin/env python3
"""VGPR liveness over a gfx9 assembly listing (one kernel): reports the program points of highest pressure and which registers are
live there.  Approximate (every write kills; partial-exec writes are treated as full), good enough to see what a spill is made of."""
import re, sys
src, kern = sys.argv[1], sys.argv[2]
L = open(src).read().split('\n')
beg = next(i for i, l in enumerate(L) if l.startswith(kern) and re.match(r'\S+:', l))
end = next(i for i in range(beg, len(L)) if 's_endpgm' in L[i])
body = L[beg:end + 1]
def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok): out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r'\bv(\d+)\b', tok))
    return out
NODEST = ('ds_write', 'buffer_store', 'global_store', 'scratch_store', 'flat_store', 'v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop', 'global_atomic', 'buffer_atomic', 'ds_add', 's_')
RMW = ('v_writelane', 'v_mac', 'v_fmac')
blocks, cur = [], {'label': 'entry', 'ins': [], 'succ': [], 'start': beg}
labels = {}
for i, l in enumerate(body):
    m = re.match(r'(\.LBB\d+_\d+):', l)
    if m:
        nb = {'label': m.group(1), 'ins': [], 'succ': [], 'start': beg + i}
        if cur.get('fall', True): cur['succ'].append(m.group(1))
        blocks.append(cur); cur = nb; continue
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') or t.startswith(';'): continue
    parts = t.split(None, 1); op = parts[0]; ops = parts[1] if len(parts) > 1 else ''
    if op.startswith('s_cbranch'): cur['succ'].append(ops.strip())
    elif op == 's_branch': cur['succ'].append(ops.strip()); cur['fall'] = False
    elif op == 's_endpgm': cur['fall'] = False
    elif op.startswith(('v_', 'ds_', 'buffer_', 'global_', 'scratch_', 'flat_')):
        o = [x.strip() for x in ops.split(',')]
        if op.startswith(NODEST): d, u = set(), set().union(*[regs(x) for x in o]) if o else set()
        else:
            d = regs(o[0]); u = set().union(*[regs(x) for x in o[1:]]) if len(o) > 1 else set()
            if op.startswith(RMW) or 'dst_unused:UNUSED_PRESERVE' in t or ('_dpp' in op and 'bound_ctrl' not in t) or 'row_' in t and 'bound_ctrl' not in t: u |= d
        cur['ins'].append((beg + i, d, u, t))
blocks.append(cur)
byl = {b['label']: b for b in blocks}
for b in blocks:
    use, df = set(), set()
    for _, d, u, _t in b['ins']:
        use |= (u - df); df |= d
    b['use'], b['def'], b['in'], b['out'] = use, df, set(), set()
ch = True
while ch:
    ch = False
    for b in reversed(blocks):
        out = set().union(*[byl[s]['in'] for s in b['succ'] if s in byl]) if b['succ'] else set()
        inn = b['use'] | (out - b['def'])
        if out != b['out'] or inn != b['in']: b['out'], b['in'], ch = out, inn, True
pts = []
for b in blocks:
    live = set(b['out'])
    for ln, d, u, t in reversed(b['ins']):
        live = (live - d) | u
        pts.append((len(live), ln, t, frozenset(live)))
pts.sort(key=lambda x: -x[0])
print('max pressure', pts[0][0], 'at line', pts[0][1] + 1, pts[0][2])
seen = 0
for n, ln, t, live in pts[:int(sys.argv[3]) if len(sys.argv) > 3 else 5]:
    print(n, ln + 1, t)
want = int(sys.argv[4]) if len(sys.argv) > 4 else None
if want:
    for n, ln, t, live in pts:
        if ln + 1 == want: print('live at', want, n, sorted(live)); break
# loop-invariant registers: live at the asked point and never written between the two given lines (argv[5], argv[6])
if len(sys.argv) > 6:
    lo, hi = int(sys.argv[5]), int(sys.argv[6])
    wr = set()
    for b in blocks:
        for ln, d, u, t in b['ins']:
            if lo <= ln + 1 <= hi: wr |= d
    for n, ln, t, live in pts:
        if ln + 1 == want:
            inv = sorted(live - wr); print('never written in', lo, hi, ':', len(inv), inv)
            break
import os
for lab in os.environ.get('LABELS', '').split():
    b = byl[lab]; print('live-in', lab, len(b['in']), sorted(b['in']))
if os.environ.get('FIRSTUSE'):
    lab = os.environ['FIRSTUSE']; b0 = byl[lab]
    first = {}
    for b in blocks:
        for ln, d, u, t in b['ins']:
            if ln >= b0['start']:
                for r in u:
                    if r in b0['in'] and (r not in first or ln < first[r][0]): first[r] = (ln + 1, t)
    hist = {}
    for r, (ln, t) in sorted(first.items(), key=lambda x: x[1][0]): hist.setdefault(ln // 250 * 250, []).append(r)
    for k in sorted(hist): print(k, len(hist[k]), hist[k])
