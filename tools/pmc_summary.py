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
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'SQ_BUSY_CYCLES' in cs and sum(cs['SQ_BUSY_CYCLES']) > 0:
        # matrix-pipe duty: MFMA-busy cycles summed over the SIMDs of a counter instance / (busy cycles x 32 SIMDs per instance) -- the
        # normalisation of profiles/r03_sq_counters.txt (calibrated there on igemm2<1,4,2>: 0.78 at 125 of 157 TFLOP/s)
        print('    %-28s %.3f' % ('=> matrix-pipe duty', sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / (32.0 * sum(cs['SQ_BUSY_CYCLES']))))
