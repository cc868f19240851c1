#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 120 python -m pytest tests/test_gpu_robust.py -m gpu -q -x 2>&1 | tail -2
