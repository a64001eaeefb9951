#!/usr/bin/env python3
"""Kernel resource table of a HIP source: tools/kres.py file.hip [name-filter] (VGPR incl. AGPR, SGPR, scratch, LDS)."""
import re, subprocess, sys, os, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + '.s')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                       '-I' + os.path.dirname(os.path.abspath(src)), '-o', out, src], stderr=subprocess.DEVNULL)
txt = open(out).read()
meta = txt[txt.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - \.agpr_count:', '\n' + meta)[1:]:
    blk = '.agpr_count:' + blk
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = g('name')
    if flt and flt not in name:
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0]
    print('%-44s vgpr %3s agpr %3s sgpr %3s scratch %4s lds %6s spill %s' % (
        dem[-44:], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('private_segment_fixed_size'),
        g('group_segment_fixed_size'), g('vgpr_spill_count')))
print('asm:', out)
