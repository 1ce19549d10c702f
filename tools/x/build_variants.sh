#!/bin/bash
# Kernel library built from the SAME sources with other AMDGPU machine-scheduler strategies -> tools/x/variants/libzkp_<name>.so (git-ignored; they
# travel to the GPU box with the snapshot).  tools/x/exp_schedules.sh swaps them in turn under bench.py.  Resource usage per kernel: build_<name>.log.
cd "$(dirname "$0")/variants" 2>/dev/null || { mkdir -p "$(dirname "$0")/variants" && cd "$(dirname "$0")/variants"; }
SRC=../../../zkp_amd/csrc/zkp_kernels.hip
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Rpass-analysis=kernel-resource-usage"
hipcc $FLAGS $SRC -o libzkp_base.so > build_base.log 2>&1 &
for v in max-ilp; do        # (iterative-ilp: the compiler of ROCm 7.2.0 crashes on this file)
  hipcc $FLAGS -mllvm -amdgpu-sched-strategy=$v $SRC -o libzkp_$v.so > build_$v.log 2>&1 &
done
wait
