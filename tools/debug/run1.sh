cd /root/repo
bash tools/gpu_kstats.sh bn tools/debug/bn_sweep.py 2>&1 | head -7 | cut -c1-150
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "batchnorm or norm or leaky or concat or epilogue or deform or loss" 2>&1 | tail -3
python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_d.json 2>/dev/null; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_d.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'])
for k,v in sorted(d['roofline']['families'].items(), key=lambda kv:-kv[1]['ms_per_step']): print('%-16s %7.2f ms %s' % (k, v['ms_per_step'], v.get('algorithmic_gbs','')))
PY
