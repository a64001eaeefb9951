"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_summary.py <csv> [name-filter ...]"""
import collections, csv, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
flt = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if flt and not any(f in name for f in flt):
        continue
    rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in sorted(rows.items()):
    n = max(len(v) for v in cs.values())
    print('%-40s launches %d' % (name[:40], n))
    for c, v in sorted(cs.items()):
        print('    %-28s %.4g' % (c, sum(v) / len(v)))
