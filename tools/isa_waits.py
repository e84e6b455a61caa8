"""ISA lint: per kernel of a .s file, the sequence of global loads (L), stores (S), vmcnt waits (Wn), barriers (B) and
MFMA groups (M) -- a `W0` right behind a group of loads inside a loop is a prefetch the compiler serialised."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w+):', txt, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    b = txt.index('s_endpgm', m.end())
    ev = []
    for l in txt[m.end():b].split('\n'):
        t = l.strip()
        if t.startswith('global_load') or t.startswith('buffer_load'): ev.append('L')
        elif t.startswith('global_store') or t.startswith('global_atomic'): ev.append('S')
        elif 'vmcnt' in t: ev.append('W' + re.search(r'vmcnt\((\d+)\)', t).group(1))
        elif t.startswith('s_barrier'): ev.append('B')
        elif 'v_mfma' in t:
            if not ev or ev[-1] != 'M': ev.append('M')
    print(name[:80]); print('  ' + ' '.join(ev)[:int(sys.argv[3]) if len(sys.argv) > 3 else 1500])
