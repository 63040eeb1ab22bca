#!/bin/bash
# Variants of the two-waves-per-SIMD policy kernel experiment (tools/pf_abl/policy8.hip), each built into its own shared library
# next to this script; tools/pf_abl/run8.py times them and checks their logits against the product kernel bit for bit,
# tools/pf_abl/stamps8.py prints the in-kernel stamps of libpf8_stamps.so.  Not part of the product build.
set -e
cd "$(dirname "$0")/../.."
for v in f32:"" f32swap:"-DPF8_SWAP" pl:"-DPF8_W_PLANES" plswap:"-DPF8_W_PLANES -DPF8_SWAP" plswap_l1:"-DPF8_W_PLANES -DPF8_SWAP -DPF_ABL_W_L1 -DPF_ABL_X_L1" \
         stamps:"-DPF8_W_PLANES -DPF8_SWAP -DPF8_STAMPS" $EXTRA; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPF_ONLY_43 $flags -Iinclude tools/pf_abl/policy8.hip -o tools/pf_abl/libpf8_$name.so &
done
wait
ls -la tools/pf_abl/libpf8_*.so
