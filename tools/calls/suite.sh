#!/bin/bash
# the whole GPU suite with the [parity] / [drift] lines kept (profiles/r05_parity.txt) and the wall time
set -u
O=gpurun_out/suite; rm -rf $O; mkdir -p $O
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -s -p no:cacheprovider > $O/pytest_full.txt 2>&1
echo "exit $? wall $(( $(date +%s) - T0 )) s" >> $O/pytest_full.txt
sed -n "s/^[.sFE]*\\(\\[\\(parity\\|drift\\|smoke\\)\\]\\)/\\1/p" $O/pytest_full.txt > $O/parity.txt      # (pytest prints its progress dots in front of a test's first line)
tail -3 $O/pytest_full.txt; wc -l $O/parity.txt
