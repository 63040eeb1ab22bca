"""Build recipe for the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m hopwise_amd.build            # builds hopwise_amd/csrc/libhopwise_amd.so in-tree

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""

import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libhopwise_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "..", "..", "include", "*.h"))
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


TORCH_LIB = os.path.join(CSRC, "libhopwise_amd_torch.so")


def build_torch_ops(force=False, verbose=True):
    """csrc/torch_bind.cpp -> libhopwise_amd_torch.so: the TORCH_LIBRARY(hopwise_amd) registration of the C ABI, compiled
    against the installed torch's headers and linked to libhopwise_amd.so (found next to it through $ORIGIN)."""
    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(CSRC, "torch_bind.cpp")
    hdrs = glob.glob(os.path.join(CSRC, "..", "..", "include", "*.h"))
    build(force=False, verbose=verbose)
    if not (force or _stale(TORCH_LIB, [src, LIB] + hdrs)):
        return TORCH_LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-unused-result", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include"]
    cmd += [src, "-o", TORCH_LIB, f"-L{CSRC}", "-lhopwise_amd", "-Wl,-rpath,$ORIGIN", f"-L{tlib}", "-ltorch", "-ltorch_cpu",
            "-ltorch_hip", "-lc10", "-lc10_hip", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
    print(build_torch_ops(force="--force" in sys.argv))
