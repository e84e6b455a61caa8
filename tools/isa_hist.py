"""ISA histogram: instruction classes (VALU / packed / transcendental / MFMA / LDS / VMEM / SALU / waits) of a kernel in a
.s file (hipcc -S --cuda-device-only), for the whole kernel and for every backward-branch loop; with a third argument
"a-b" the opcode histogram of that line range.  python tools/isa_hist.py fused.s <mangled kernel name> [first-last]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
name = sys.argv[2]
m = re.search(r'^%s:' % re.escape(name), txt, re.M)
b = txt.index('s_endpgm', m.end())
lines = [l.strip() for l in txt[m.end():b].split('\n')]
# find labels and backward branches
lab = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: lab[mm.group(1)] = i
loops = []
for i, l in enumerate(lines):
    mm = re.match(r'^s_cbranch_\w+ (\.LBB\d+_\d+)|^s_branch (\.LBB\d+_\d+)', l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in lab and lab[t] < i: loops.append((lab[t], i))
def cls(l):
    op = l.split()[0] if l else ''
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')): return 'trans'
    if op.startswith('v_pk_'): return 'valu_pk'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return None
def hist(a, b):
    c = collections.Counter()
    for l in lines[a:b + 1]:
        k = cls(l)
        if k: c[k] += 1
    return dict(c)
print('whole', hist(0, len(lines) - 1))
for a, b in sorted(loops, key=lambda t: t[0]):
    if b - a > 30: print('loop lines %d-%d' % (a, b), hist(a, b))
if len(sys.argv) > 3:
    a, b = map(int, sys.argv[3].split('-'))
    c = collections.Counter(l.split()[0] for l in lines[a:b+1] if cls(l) in ('valu','valu_pk','trans'))
    for k, v in c.most_common(40): print('  %-28s %d' % (k, v))
