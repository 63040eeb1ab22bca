#!/bin/bash
# A/B of the whole 3-stream PGPR search with the product policy kernel against the experiment forms (built into
# tools/pf_abl/lib_*.alt): alternates the libraries in one box session so clock drift hits both alike.
set -e
cd "$GRAFT_REPO_ROOT"
L=hopwise_amd/csrc/libhopwise_amd.so
cp $L /tmp/lib_base.so
out=gpurun_out/ab_streams.txt; : > $out
for rep in 1 2 3; do
  for v in base glds pipe; do
    if [ $v = base ]; then cp /tmp/lib_base.so $L; else cp tools/pf_abl/lib_$v.alt $L; fi
    for s in 1 3; do
      echo "== $v streams=$s rep=$rep" >> $out
      python bench_pgpr.py --users 256 --streams $s --steps 0 2>>gpurun_out/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out
    done
  done
done
cp /tmp/lib_base.so $L
cat $out
