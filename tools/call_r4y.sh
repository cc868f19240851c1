#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q > $O/r4y_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4y_pytest.log
tail -6 $O/r4y_pytest.log
