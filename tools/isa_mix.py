"""Developer aid: instruction mix per basic block of a kernel dumped by tools/isa_report.py --dump."""
import re, collections, sys
body = open(sys.argv[1]).read().split('\n')
minlen = int(sys.argv[2]) if len(sys.argv) > 2 else 60
blocks = []; cur = ['entry', []]; blocks.append(cur)
for l in body:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t): cur = [t, []]; blocks.append(cur)
    elif t and not t.startswith(';') and not t.startswith('.'):
        cur[1].append(t.split()[0])
for b in blocks:
    c = collections.Counter(b[1]); cat = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_pk'): cat['vpk'] += v
        elif k.startswith(('v_sqrt', 'v_rcp', 'v_rsq')): cat['trans'] += v
        elif k.startswith('v_'): cat['valu'] += v
        elif k.startswith('s_'): cat['salu'] += v
        elif k.startswith('ds_'): cat['lds'] += v
        elif k.startswith('buffer'): cat['buf'] += v
        else: cat[k] += v
    if len(b[1]) >= minlen:
        print(b[0][:40], len(b[1]), dict(cat))
        print('   ', sorted(c.items(), key=lambda x: -x[1])[:24])
