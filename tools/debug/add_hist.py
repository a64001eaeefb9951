"""Histogram of the torch elementwise-add launches of one profiled bench run: python tools/debug/add_hist.py <kernel_trace.csv>"""
import csv, sys, collections
h = collections.Counter(); tm = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'at::native' not in n:
        continue
    key = ('add' if 'CUDAFunctor_add' in n else n.split('<')[0][-40:], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0)))
    h[key] += 1
    tm[key] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for k, t in sorted(tm.items(), key=lambda kv: -kv[1])[:25]:
    print('%-44s grid %10d  launches %5d  total %8.3f ms' % (k[0], k[1], h[k], t))
