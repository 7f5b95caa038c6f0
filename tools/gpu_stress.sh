#!/bin/bash
for i in 1 2 3; do ( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 ); done
timeout 300 python -c "
import __graft_entry__ as g; g.smoke()"
