cd /root/repo
timeout 900 python tools/debug/dcn_bwd_check.py parity time 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "side_stream or reference_fixture or flat_grad" 2>&1 | tail -3
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_b.json 2>/dev/null; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_b.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k,v in sorted(d['roofline']['families'].items(), key=lambda kv:-kv[1]['ms_per_step']): print('%-16s %7.2f ms' % (k, v['ms_per_step']))
PY
