#!/bin/bash
# tools/scorer_ab.sh <macro>[=value]: bench.py's headline leg with the library as built (A), then with kge_score*.hip rebuilt on
# the box with -D<macro>=<value> (default 0) (B), then A again -- same card, back to back.
MACRO=${1:-HW_ROW_BUFFER_LOAD}
case "$MACRO" in *=*) ;; *) MACRO="$MACRO=0" ;; esac
mkdir -p gpurun_out/scorer_ab
ARGS="--no-pgpr --no-tucker --no-cafe --no-sharded --no-cpu-baseline"
run() { timeout -k 10 300 python bench.py $ARGS > gpurun_out/scorer_ab/$1.json 2> gpurun_out/scorer_ab/$1.err; python - <<PY
import json
d=json.loads(open('gpurun_out/scorer_ab/$1.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$1', 'G triples/s', round(d['value']/1e9,3), 'us/batch', round(d['config']['us_per_batch'],2), 'hbm fused us', round(r['kernel_ms']*1e3,1), 'plain', round(r['plain_scorer']['kernel_ms']*1e3,1), 'configs1 us', round(r['configs1_launch']['kernel_ms']*1e3,1))
PY
}
run A1
cd hopwise_amd/csrc
for f in kge_score kge_score_ext; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -D$MACRO -c $f.hip -o /tmp/$f.o || exit 1; done
cp libhopwise_amd.so /tmp/lib_a.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhopwise_amd.so $(ls *.o | grep -v "^kge_score.o$" | grep -v "^kge_score_ext.o$") /tmp/kge_score.o /tmp/kge_score_ext.o || exit 1
cd ../..
run B1
cp /tmp/lib_a.so hopwise_amd/csrc/libhopwise_amd.so
run A2
