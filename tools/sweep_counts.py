#!/usr/bin/env python3
"""Static instruction counts of the unrolled group blocks of fill_regs.hip (markers are added to a scratch copy of the source)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(root, 'stitch_amd/csrc/fill_regs.hip')).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep('#define P1(g) if (GUARD(g)) { uint32_t tbw; \\', '#define P1(g) if (GUARD(g)) { uint32_t tbw; asm volatile("; P1G_BEGIN"); \\')
rep('            tb_lane[(g) * 64] = tbw; }\n        REP20(P1)', '            tb_lane[(g) * 64] = tbw; asm volatile("; P1G_END"); }\n        REP20(P1)')
rep('#define P1B(g) if (GUARD(g)) { \\', '#define P1B(g) if (GUARD(g)) { asm volatile("; P1BG_BEGIN"); \\')
rep('            tb_lane[(g) * 64] = tbw; }\n        REP20(P1B)', '            tb_lane[(g) * 64] = tbw; asm volatile("; P1BG_END"); }\n        REP20(P1B)')
rep('                P2TAIL(g) \\', '                asm volatile("; TAIL_BEGIN"); P2TAIL(g) asm volatile("; TAIL_END"); \\')
tmp = os.path.join(root, 'stitch_amd/csrc/_m.hip')
open(tmp, 'w').write(s)
try:
    subprocess.run(['/opt/rocm/bin/hipcc', '-w', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-mllvm', '-amdgpu-sched-strategy=iterative-ilp', tmp, '-o', '/tmp/m.s'], check=True)
finally:
    os.remove(tmp)
L = open('/tmp/m.s').read().split('\n')
end = next(i for i, l in enumerate(L) if 's_endpgm' in l)
for tag in ['P1G', 'P1BG', 'TAIL']:
    res = []; i = 0
    while i < end:
        if '; %s_BEGIN' % tag in L[i]:
            j = i
            while '; %s_END' % tag not in L[j] and j < end: j += 1
            res.append((sum(1 for l in L[i:j] if re.match(r'\s+v_', l)), sum(1 for l in L[i:j] if re.match(r'\s+s_', l)), sum(1 for l in L[i:j] if 's_cbranch' in l), sum(1 for l in L[i:j] if 'v_mov_b32' in l), sum(1 for l in L[i:j] if 'v_readlane' in l), sum(1 for l in L[i:j] if 's_nop' in l), sum(1 for l in L[i:j] if 's_waitcnt' in l))); i = j
        i += 1
    print(tag, len(res), '(valu, salu, branches, v_mov, v_readlane, s_nop, s_waitcnt) per group:', res[:10])
