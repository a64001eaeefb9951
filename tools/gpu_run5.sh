#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --shapes gpurun_out/shapes_r2a.txt > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err
cat gpurun_out/bench_r2a.json
head -70 gpurun_out/shapes_r2a.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
