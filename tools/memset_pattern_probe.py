"""Does a replayed memset NODE write the value it was captured with -- whatever the process does between the replays?

Round 4, DESIGN.md §2.  tools/graph_dump.py showed that the capture of the training step is one linear chain in which the
memset node has its predecessor and its successor like every kernel node, and that no other tensor of the capture ever
owned the memset's destination block.  So neither a missing edge nor an aliased pool block explains the garbage the
replica scratch held in round 3 (16-byte records, the first 8 bytes wrong, not the NaN poison).  What is left is how the
runtime EXECUTES a memset node.  One candidate this probe can decide in a single run, without waiting for a rare event: a
memset is a fill kernel of the runtime whose pattern lives in a staging buffer of the runtime, not in the graph; if the
instantiated graph keeps pointing at a staging slot that later eager fills / copies of the same process recycle, a replay
writes THEIR pattern.

The capture: one hipMemsetAsync(dst, 0, 51 200 bytes) (the replica scratch's size).  Between replays the process issues
eager hipMemsetAsync calls with other byte values on another buffer, device-to-host reads and device-to-device copies (what
a training loop does: loss.item(), torch.zeros, ...).  Before each replay dst is set to 0x55 by a kernel; after it every
byte must be 0.  Prints the first replay at which it is not, and what dst held."""
import argparse
import ctypes as C
import json
import os

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=400)
ap.add_argument("--eager-memsets", type=int, default=64, help="eager hipMemsetAsync calls between two replays")
ap.add_argument("--bytes", type=int, default=51200)
ap.add_argument("--chain", action="store_true",
                help="capture [kernel: dst <- 0x55] -> [memset dst 0] -> [kernel: snap <- dst] instead of the lone memset: the "
                     "shape of the training step's capture (a memset node between kernel nodes), checked on snap")
ap.add_argument("--out", default="")
args = ap.parse_args()

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
dev = torch.device("cuda", 0)
dst = torch.empty(args.bytes, dtype=torch.uint8, device=dev)
other = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
src = torch.arange(1 << 18, dtype=torch.int32, device=dev)
cpy = torch.empty_like(src)
torch.cuda.synchronize()

snap = torch.empty_like(dst)
if args.chain:  # warm the two torch kernels up outside the capture
    dst.fill_(0x55)
    snap.copy_(dst)
    torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if args.chain:
        dst.fill_(0x55)
    rc = hip.hipMemsetAsync(dst.data_ptr(), 0, args.bytes, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    if args.chain:
        snap.copy_(dst)

result = {"capture": "kernel -> memset -> kernel" if args.chain else "memset alone", "replays": args.replays, "eager_memsets_between_replays": args.eager_memsets, "bytes": args.bytes,
          "first_bad_replay": None, "bad_replays": 0}
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
n_eager = 0
for r in range(args.replays):
    for j in range(args.eager_memsets):  # eager fills with other patterns, odd sizes and offsets included
        val = 1 + (n_eager % 250)
        size = (4096, 51200, 777, 1 << 16, 12)[n_eager % 5]
        off = (0, 3, 64, 1001)[n_eager % 4]
        assert hip.hipMemsetAsync(other.data_ptr() + off, val, size, stream) == 0
        n_eager += 1
    cpy.copy_(src)  # a device-to-device copy
    _ = int(src[r % src.numel()].item())  # a device-to-host read (what loss.item() is)
    dst.fill_(0x55)  # a kernel: the replay must overwrite all of it
    snap.fill_(0x33)
    g.replay()
    torch.cuda.synchronize()
    seen = snap if args.chain else dst
    bad = seen != 0
    if bool(bad.any()):
        result["bad_replays"] += 1
        if result["first_bad_replay"] is None:
            idx = bad.nonzero().flatten()
            vals, counts = seen[bad].unique(return_counts=True)
            result["first_bad_replay"] = r
            result["eager_memsets_before_it"] = n_eager
            result["bad_bytes"] = int(idx.numel())
            result["bad_byte_range"] = [int(idx.min()), int(idx.max())]
            result["bad_values"] = {int(v): int(c) for v, c in zip(vals.tolist()[:16], counts.tolist()[:16])}
            result["first_64_bytes_hex"] = bytes(seen[:64].cpu().tolist()).hex()
print(json.dumps(result))
if args.out:
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(result, f, indent=1)
