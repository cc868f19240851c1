#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 50 python -m pytest tests/test_gpu_random.py -m gpu -q -x -k "rows_exp_ahead" 2>&1 | tail -1; done
