"""Histogram of instruction classes along one kernel of a hipcc -save-temps .s file."""
import sys
from collections import Counter

path, name = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 500
S = open(path).read()
k = S.split(name + ':')[1].split('.end_amdhsa_kernel')[0]
ins = []
for l in k.split('\n'):
    t = l.strip()
    if not l.startswith('\t') or not t or t[0] in '.;':
        continue
    ins.append(t.split()[0])
print(len(ins), 'instructions')
for i in range(0, len(ins), step):
    c = Counter()
    for x in ins[i:i + step]:
        if x.startswith('scratch_store'): c['sst'] += 1
        elif x.startswith('scratch_load'): c['sld'] += 1
        elif x.startswith('v_accvgpr'): c['acc'] += 1
        elif x.startswith('v_fma_f64'): c['fma'] += 1
        elif x in ('v_min_f64', 'v_max_f64'): c['mm'] += 1
        elif x.startswith('ds_'): c['ds'] += 1
        elif x.startswith('global_'): c['gl'] += 1
        elif x.startswith('v_'): c['v'] += 1
        elif x.startswith('s_waitcnt'): c['wait'] += 1
        elif x.startswith('s_'): c['s'] += 1
    print(i, dict(sorted(c.items())))
