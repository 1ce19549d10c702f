#!/bin/bash
# scratch job script for gpurun (edited per call)
set -u
export TMPDIR=/tmp
O=gpurun_out
TAG=${1:-r02i}
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --durations=0 > $O/${TAG}_pytest_gpu_durations.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest_gpu_durations.txt
grep -E "passed|failed|rc=|Error" $O/${TAG}_pytest_gpu_durations.txt | tail -5
