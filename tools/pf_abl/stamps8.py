#!/usr/bin/env python
"""In-kernel s_memtime stamps of the eight waves of one block of the two-waves-per-SIMD policy kernel experiment
(libpf8_stamps.so: -DPF8_STAMPS): per hidden-layer trip, four stamps per wave -- trip start, end of the first phase, end of the
second, arrival at the barrier.  Half 0 (waves 0-3) prepares operands first and multiplies afterwards in the PF8_SWAP build,
half 1 the other way round."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import run8

lib, product, experiment, keep = run8.load(os.path.join(HERE, sys.argv[1] if len(sys.argv) > 1 else "libpf8_stamps.so"))
out = torch.empty((run8.rows, 256), device=run8.dev)
for _ in range(3):
    experiment(out)
torch.cuda.synchronize()
st = np.zeros(8 * 256, np.uint64)
assert lib.pf8_read_stamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.reshape(8, 256)[:, : 16 * 4].astype(np.int64)
st = (st - st[:, 0].min()).reshape(8, 16, 4)
print("wave (half): per trip [start, phase-1 end, phase-2 end, at barrier], cycles since the block's first stamp")
for w in (0, 4, 1, 5):
    print("wave", w, "half", w >> 2)
    for c in range(16):
        print("   trip %2d  " % c + "  ".join("%7d" % v for v in st[w, c]),
              "  | trip length", st[w, c + 1, 0] - st[w, c, 0] if c < 15 else "")
