#!/bin/bash
# Round 5, the collection call: the GPU suite and the collection of every profile and bench line on the final sources (row-limb Horner tail and inversion,
# whole windows per XCD in the scatter pass); the constant-time evidence and the host-scaling table of tools/x/r05_job13.sh stay (term kernel and pipe unchanged).
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out
mkdir -p $O
echo "== 1. GPU suite"
python -m pytest tests -m gpu -x -q --durations=15 2>&1 | tail -25 > $O/r05_pytest_gpu_durations.txt; tail -22 $O/r05_pytest_gpu_durations.txt
echo "== 2. collection"
bash tools/final_collection.sh 2>&1 | tail -30
echo "== 3. soak"
timeout 100 python tools/pipe_soak.py 30 1717 2>&1 | tail -2
timeout 100 python tools/soak.py 30 1717 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
