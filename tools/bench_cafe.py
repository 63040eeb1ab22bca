#!/usr/bin/env python
"""CAFE run_program on the ml-1m-small-shaped CKG: the `cafe` leg of bench.py by itself (bench_cafe.py at the repo root).
    python tools/bench_cafe.py [--users 256] [--metapaths 6] [--cpu-users 0]"""
import os
import runpy
import sys

sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_cafe.py")
runpy.run_path(sys.argv[0], run_name="__main__")
