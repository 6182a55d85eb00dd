#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
time (timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3)
