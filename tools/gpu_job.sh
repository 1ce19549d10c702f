#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
grep -E "passed|failed|rc=|Error|assert" $O/${TAG}_pytest.txt | tail -8
