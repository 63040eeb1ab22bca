#!/bin/bash
# Timing-only variants of the fused PGPR policy kernel (csrc/pgpr_policy.hip, PF_ABL_* switches): each built into its own
# shared library next to this script; tools/pf_abl/run.py times them on synthetic beam rows.  Not part of the product build.
set -e
cd "$(dirname "$0")/../.."
for v in base:"" x_single:"-DPF_SINGLE_X=1" x_glds:"-DPF_GLDS=1" x_glds3:"-DPF_GLDS=1 -DPF_NSTAGE=3" x_pipe:"-DPF_PIPE=1" x_pipe11:"-DPF_PIPE=1 -DPF_XPS=1 -DPF_X0=1" p_xl1:"-DPF_PIPE=1 -DPF_ABL_X_L1" nomma:"-DPF_ABL_NO_MMA" nox:"-DPF_ABL_NO_X" now:"-DPF_ABL_NO_W -DPF_ABL_NO_SYNC" nosync:"-DPF_ABL_NO_SYNC" mmaonly:"-DPF_ABL_NO_X -DPF_ABL_NO_W -DPF_ABL_NO_SYNC"; do
    name=${v%%:*}; flags=${v#*:}
    src=hopwise_amd/csrc/pgpr_policy.hip
    case $name in x_*|p_*) src=tools/pf_abl/policy_pipe.hip;; esac  # round-5 experiment forms (bit-identical to the product kernel)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPF_ONLY_43 $flags -Iinclude $src -o tools/pf_abl/libpf_$name.so &
done
wait
ls -la tools/pf_abl/*.so
