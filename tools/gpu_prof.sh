#!/bin/bash
# rocprofv3 kernel stats of the bench step: tools/gpu_prof.sh <tag> [bench args]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}"
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-detail --wgrad-inline --no-graph "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv under $out (rocprofv3 failed, see the .log beside it)" >&2; exit 1; }
cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
fam = collections.OrderedDict()
def family(n):
    for key, lab in (('pointwise_wgrad', 'conv pointwise'), ('pointwise_kernel', 'conv pointwise'), ('igemm3_pack', 'weight pack'), ('x9_wexp', 'weight pack'), ('igemm3_x9', 'conv fwd/dgrad (x9)'), ('igemm2_tr2', 'conv fwd/dgrad (igemm2)'), ('igemm2_kernel', 'conv fwd/dgrad (igemm2)'), ('igemm2_pack', 'weight pack'), ('conv_igemm_kernel', 'conv fwd/dgrad (gen1)'),
                     ('wgrad2_kernel', 'conv wgrad (wgrad2)'), ('wgrad2_reduce', 'conv wgrad (wgrad2)'), ('conv_wgrad_kernel', 'conv wgrad (gen1)'), ('repack', 'weight pack'),
                     ('dcn_lean_fwd', 'dcn fwd'), ('dcn_fwd', 'dcn fwd'), ('dcn_lean_bwd_offset', 'dcn offset+wgrad'), ('dcn_bwd_offset', 'dcn offset+wgrad'), ('dcn_bwd_input', 'dcn input'), ('dcn_', 'dcn misc'),
                     ('bn_', 'norm/act'), ('smallk', 'smallk conv'), ('at::native', 'torch elementwise'), ('rocclr', 'fill/copy'), ('copy_channels', 'layout'), ('swap_axes', 'layout'),
                     ('head_', 'softargmin head'), ('dw_', 'depthwise'), ('bilinear', 'resample'), ('nearest', 'resample')):
        if key in n:
            return lab
    return 'other'
tot = 0.0
for r in rows:
    t = float(r['TotalDurationNs']) / 4e6      # 4 steps (1 warm-up + 3 timed)
    f = fam.setdefault(family(r['Name']), [0.0, 0])
    f[0] += t; f[1] += int(r['Calls'])
    tot += t
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print('%-28s %8.2f ms/step  %6d launches/step' % (k, t, c // 4))
print('%-28s %8.2f ms/step' % ('TOTAL kernel time', tot))
PY
