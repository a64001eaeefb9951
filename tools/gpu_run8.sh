#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|max err" | head -8; done
