#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_e2e.py -x -q -k "c2_shape or reference_fixture or no_worse" 2>&1 | tail -15
