#!/bin/bash
# tools/replay_ab.sh <processes per arm> [steps]: the graph-replay stress (tools/ref_layout_stress.py --graph --poison) in fresh
# processes, alternating between the library's kernel fills and HW_FILL_MEMSET=1 (hipMemsetAsync, i.e. memset nodes in the
# capture) on the same tree; one line per process in gpurun_out/replay_ab.log, the tally at the end.
N=${1:-20}; STEPS=${2:-120}
mkdir -p gpurun_out; : > gpurun_out/replay_ab.log
for i in $(seq 1 $N); do
  for arm in 0 1; do
    out=$(HW_FILL_MEMSET=$arm timeout -k 5 120 python tools/ref_layout_stress.py --steps $STEPS --graph --poison 2>&1 | grep -E "^ok|^step" | head -1)
    echo "memset=$arm run=$i :: $out" >> gpurun_out/replay_ab.log
  done
done
for arm in 0 1; do
  echo "memset=$arm: $(grep -c "memset=$arm .*:: ok" gpurun_out/replay_ab.log) ok, $(grep -c "memset=$arm .*:: step" gpurun_out/replay_ab.log) failed"
done
