"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel-family HBM traffic per launch.

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json] [train steps in each pass]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (MI355X_MICROARCH.md: hbm_bytes = (FETCH + WRITE) * 1024); on gfx950
FETCH_SIZE under-counts wide coalesced reads by 2x -- the raw value is kept and the corrected one (x2) reported beside it.
"""
import collections
import csv
import json
import sys


def family(name):
    for key, fam in (('igemm2', 'igemm2'), ('igemm3', 'igemm2'), ('wgrad2', 'wgrad2'), ('pointwise', 'pointwise'), ('conv_igemm_kernel', 'conv_igemm_kernel'),
                     ('conv_wgrad_kernel', 'conv_wgrad_kernel'), ('dcn_bwd_input', 'dcn_bwd_input'), ('dcn_lean_fwd', 'dcn_fwd'), ('dcn_fwd', 'dcn_fwd'),
                     ('dcn_lean_bwd_offset', 'dcn_bwd_offset'), ('dcn_bwd_offset', 'dcn_bwd_offset'), ('dcn_wgrad_fold', 'dcn_bwd_offset'),
                     ('smallk', 'smallk'), ('bn_', 'bn_'), ('head_', 'head_')):
        if key in name:
            return fam
    return None


def kernel_sources_sha16():
    """sha-256 (first 16 hex digits) over the CODE of the kernel sources the numbers belong to -- comments and white space are stripped first,
    so that a comment edit does not orphan a committed traffic file: bench.py recomputes it and says whether the committed
    file still matches the kernels it runs."""
    import glob
    import hashlib
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dualpixelface_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, '*.hip')) + glob.glob(os.path.join(root, '*.h'))):
        src = open(f, encoding='utf-8', errors='replace').read()
        src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
        src = re.sub(r'//[^\n]*', ' ', src)
        h.update(os.path.basename(f).encode())
        h.update(''.join(src.split()).encode())
    return h.hexdigest()[:16]


# helper launches of a family (weight packs, slab folds): their bytes belong to the operation, but "per launch" means per main kernel
HELPERS = ('igemm2_pack', 'igemm3_pack', 'wgrad2_reduce', 'wgrad2_fold', 'pointwise_wgrad_fold', 'pointwise_wgrad_reduce', 'bn_finalize', 'bn_fold', 'dcn_wgrad_fold')


def agg(path, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        fam = family(r['Kernel_Name'])
        if fam is None or r['Counter_Name'] != counter:
            continue
        tot[fam] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            if not any(h in r['Kernel_Name'] for h in HELPERS):
                n[fam] += 1
    return tot, n


def main():
    fetch, nf = agg(sys.argv[1], 'FETCH_SIZE')
    write, nw = agg(sys.argv[2], 'WRITE_SIZE')
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
    out = {}
    for fam in sorted(set(fetch) | set(write)):
        launches = max(nf.get(fam, 0), nw.get(fam, 0), 1)
        f = fetch.get(fam, 0.0) * 1024.0 / max(nf.get(fam, 1), 1)
        w = write.get(fam, 0.0) * 1024.0 / max(nw.get(fam, 1), 1)
        out[fam] = {'launches': launches, 'fetch_bytes_per_launch_raw': f, 'fetch_bytes_per_launch_x2': 2 * f,
                    'write_bytes_per_launch': w, 'hbm_bytes_per_launch': 2 * f + w}
        if steps:      # all dispatches of the family (helpers included) per train step
            out[fam]['hbm_bytes_per_step'] = (2 * fetch.get(fam, 0.0) + write.get(fam, 0.0)) * 1024.0 / steps
            out[fam]['launches_per_step'] = launches / steps
    import time
    out['_meta'] = {'kernel_sources_sha16': kernel_sources_sha16(), 'collected_utc': time.strftime('%Y-%m-%dT%H:%M:%SZ', time.gmtime()),
                    'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 '
                               '--no-cpu-baseline --no-detail --wgrad-inline', 'train_steps_in_each_pass': steps,
                    'units': 'FETCH_SIZE / WRITE_SIZE in KiB; FETCH doubled (gfx950 counts 64 B per 128-B request), MI355X_MICROARCH.md'}
    json.dump(out, open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout, indent=1)


if __name__ == '__main__':
    main()
