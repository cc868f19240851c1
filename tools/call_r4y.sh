#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "c4_shape_slice or c2_shape_slice" 2>&1 | tail -4
