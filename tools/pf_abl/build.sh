#!/bin/bash
# Timing-only variants of the fused PGPR policy kernel (csrc/pgpr_policy.hip, PF_ABL_* switches): each built into its own
# shared library next to this script; tools/pf_abl/run.py times them on synthetic beam rows.  Not part of the product build.
set -e
cd "$(dirname "$0")/../.."
for v in base:"" nomma:"-DPF_ABL_NO_MMA" nox:"-DPF_ABL_NO_X" now:"-DPF_ABL_NO_W -DPF_ABL_NO_SYNC" nosync:"-DPF_ABL_NO_SYNC" mmaonly:"-DPF_ABL_NO_X -DPF_ABL_NO_W -DPF_ABL_NO_SYNC"; do
    name=${v%%:*}; flags=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPF_ONLY_43 $flags -Iinclude hopwise_amd/csrc/pgpr_policy.hip -o tools/pf_abl/libpf_$name.so &
done
wait
ls -la tools/pf_abl/*.so
