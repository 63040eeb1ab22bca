#!/bin/bash
# Timing-only variants of the fused PGPR policy kernel, all from tools/pf_abl/policy_pipe.hip (the product kernel,
# csrc/pgpr_policy.hip, is its -DPF_HALFX=1 -DPF_BWAIT form with the switches resolved): each built into its own shared library
# next to this script; tools/pf_abl/run.py times them on synthetic beam rows.  Not part of the product build.
set -e
cd "$(dirname "$0")/../.."
for v in base:"" x_half:"-DPF_HALFX=1 -DPF_BWAIT -Wno-inline-asm" x_half_asm3:"-DPF_HALFX=1 -DPF_BWAIT -DPF_ASMDMA -DPF_NSTAGE=3 -Wno-inline-asm" x_regs:"-DPF_GLDS=0" x_single:"-DPF_GLDS=0 -DPF_SINGLE_X=1" x_glds3:"-DPF_GLDS=1 -DPF_NSTAGE=3" x_pipe:"-DPF_PIPE=1" x_pipe11:"-DPF_PIPE=1 -DPF_XPS=1 -DPF_X0=1" p_xl1:"-DPF_PIPE=1 -DPF_ABL_X_L1" nomma:"-DPF_ABL_NO_MMA" nox:"-DPF_ABL_NO_X" now:"-DPF_ABL_NO_W -DPF_ABL_NO_SYNC" nosync:"-DPF_ABL_NO_SYNC" mmaonly:"-DPF_ABL_NO_X -DPF_ABL_NO_W -DPF_ABL_NO_SYNC"; do
    name=${v%%:*}; flags=${v#*:}
    src=tools/pf_abl/policy_pipe.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPF_ONLY_43 $flags -Iinclude $src -o tools/pf_abl/libpf_$name.so &
done
wait
ls -la tools/pf_abl/*.so
