"""What does the HIP graph of a captured training step look like?  (driver.GraphedTrainStep on the bench's reference_layout
step: RotatE d = 200, the reference's 8 192-triple batch, FusedAdam.)

    python tools/graph_dump.py --out gpurun_out/graph_dump --tag NAME [--poison]

Written in round 4 for the two checks DESIGN.md §2 ("an intermittent NaN under HIP-graph replay") had left open:

  1. is a memset node of the capture connected like every other node?  The captured hipGraph_t is read back through the HIP
     graph API (hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType / hipGraphMemsetNodeGetParams via ctypes on the
     runtime torch loaded; hipGraphDebugDotPrint for the kernel names): node count by type, whether the graph is ONE linear
     chain, and the neighbours of every non-kernel node;
  2. who owned a memset's destination block before it?  torch's allocator history is recorded around the warm-up and the
     capture; every allocation of the capture's private pool that overlaps a memset destination is listed in order with
     the Python frame that asked for it.

HW_FILL_MEMSET=1 puts the library's hipMemsetAsync calls back (only in a library that still has that switch: the round-3
tree; round 4 removed it after this dump)."""
import argparse
import collections
import ctypes as C
import json
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/graph_dump")
ap.add_argument("--tag", default="capture")
ap.add_argument("--poison", action="store_true", help="torch.empty returns NaN-filled memory (a fill kernel per allocation)")
args = ap.parse_args()
os.makedirs(args.out, exist_ok=True)
dot_path = os.path.abspath(os.path.join(args.out, args.tag + ".dot"))

# the captured hipGraph_t must survive capture_end: keep_graph=True (instantiation is then explicit)
_Base = torch.cuda.CUDAGraph


class _KeptGraph(_Base):
    def __new__(cls, keep_graph=False):
        return super().__new__(cls, True)

    def __init__(self, keep_graph=False):  # pybind11 builds the C++ object here
        super().__init__(True)


torch.cuda.CUDAGraph = _KeptGraph

import hopwise_amd.model as M  # noqa: E402
from hopwise_amd.driver import GraphedTrainStep  # noqa: E402
from hopwise_amd.optim import FusedAdam  # noqa: E402
from tests.helpers import FakeDataset, make_config  # noqa: E402

if args.poison:
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
dev = torch.device("cuda", 0)
nU, nI, nE, nR, K, B, DIM = 6041, 2985, 13805, 13, 256, 2048, 200
rng = np.random.default_rng(0)
torch.manual_seed(0)
m = M.RotatE(make_config(dev, DIM, 1.0, train_neg_sample_args={"sample_num": K}), FakeDataset(nU, nI, nE, nR)).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-3)
pos_n = B // K
users, pos = rng.integers(1, nU, pos_n), rng.integers(1, nI, pos_n)
it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
    user_id=np.tile(users, K), item_id=np.tile(pos, K), neg_item_id=rng.integers(1, nI, B),
    head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
    neg_tail_id=rng.integers(1, nE, B)).items()}

torch.cuda.memory._record_memory_history(max_entries=200000, stacks="python")
g = GraphedTrainStep(m, opt, it)
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
graph = C.c_void_p(g.graph.raw_cuda_graph())
n_nodes = C.c_size_t(0)
assert hip.hipGraphGetNodes(graph, None, C.byref(n_nodes)) == 0
nodes = (C.c_void_p * n_nodes.value)()
assert hip.hipGraphGetNodes(graph, nodes, C.byref(n_nodes)) == 0
n_edges = C.c_size_t(0)
assert hip.hipGraphGetEdges(graph, None, None, C.byref(n_edges)) == 0
e_from, e_to = (C.c_void_p * max(n_edges.value, 1))(), (C.c_void_p * max(n_edges.value, 1))()
if n_edges.value:
    assert hip.hipGraphGetEdges(graph, e_from, e_to, C.byref(n_edges)) == 0
TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event", 7: "event_record",
         10: "mem_alloc", 11: "mem_free"}


class MemsetParams(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint), ("height", C.c_size_t), ("pitch", C.c_size_t),
                ("value", C.c_uint), ("width", C.c_size_t)]


ids = [int(n) for n in nodes]
order = {n: i for i, n in enumerate(ids)}  # hipGraphGetNodes order = creation (= capture) order
kind, memsets = {}, {}
for n in ids:
    t = C.c_int(-1)
    assert hip.hipGraphNodeGetType(C.c_void_p(n), C.byref(t)) == 0
    kind[n] = TYPES.get(t.value, str(t.value))
    if t.value == 2:
        p = MemsetParams()
        if hip.hipGraphMemsetNodeGetParams(C.c_void_p(n), C.byref(p)) == 0:
            memsets[n] = {"dst": int(p.dst or 0), "bytes": int(p.width) * int(p.elementSize) * max(int(p.height), 1),
                          "value": int(p.value)}
succ, pred = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(list(e_from)[: n_edges.value], list(e_to)[: n_edges.value]):
    succ[int(a)].append(int(b))
    pred[int(b)].append(int(a))
roots = [n for n in ids if not pred[n]]
leaves = [n for n in ids if not succ[n]]
linear = len(roots) == 1 and len(leaves) == 1 and all(len(succ[n]) <= 1 and len(pred[n]) <= 1 for n in ids)
# position of every node along the chain (0 = first to run)
pos_in_chain = {}
if linear:
    n, i = roots[0], 0
    while True:
        pos_in_chain[n] = i
        if not succ[n]:
            break
        n, i = succ[n][0], i + 1

# kernel names: the runtime's own DOT print (verbose), matched to nodes by their order in the file
rc = hip.hipGraphDebugDotPrint(graph, dot_path.encode(), C.c_uint(1))  # hipGraphDebugDotFlagsVerbose
names = {}
if rc == 0 and os.path.exists(dot_path):
    text = open(dot_path).read()
    labs = re.findall(r'^\s*"?[\w.]+"?\s*\[[^\]]*?label\s*=\s*"((?:[^"\\]|\\.)*)"', text, re.M)
    if len(labs) == len(ids):
        names = {n: re.sub(r"\s+", " ", lab.replace("\\n", " ").replace("\\l", " "))[:140] for n, lab in zip(ids, labs)}


def describe(n):
    s = f"#{pos_in_chain.get(n, order[n])} {kind[n]}"
    if n in memsets:
        s += f" dst=0x{memsets[n]['dst']:x} bytes={memsets[n]['bytes']} value={memsets[n]['value']}"
    if n in names:
        s += " :: " + names[n]
    return s


# allocator history: blocks of the capture's private pool that overlap a memset destination, in allocation order
def frame_of(ev):
    for fr in ev.get("frames", []):
        fn = fr.get("filename", "")
        if "/hopwise_amd/" in fn or "/tools/" in fn or "/tests/" in fn:
            return f"{os.path.basename(fn)}:{fr.get('line')} {fr.get('name')}"
    return ""


history = []
for trace in snap.get("device_traces", []):
    for ev in trace:
        if ev.get("action") in ("alloc", "free_requested", "free_completed", "free"):
            history.append({"action": ev["action"], "addr": int(ev.get("addr", 0)), "size": int(ev.get("size", 0)),
                            "stream": int(ev.get("stream", 0)), "where": frame_of(ev)})
owners = {}
for n, ms in memsets.items():
    lo, hi = ms["dst"], ms["dst"] + ms["bytes"]
    owners[n] = [h for h in history if h["addr"] < hi and h["addr"] + h["size"] > lo]

g.graph.instantiate()
for _ in range(3):
    loss = g(it)
torch.cuda.synchronize()

report = {"tag": args.tag, "HW_FILL_MEMSET": os.environ.get("HW_FILL_MEMSET", ""), "poison": args.poison,
          "nodes": len(ids), "edges": int(n_edges.value), "by_kind": dict(collections.Counter(kind.values())),
          "roots": len(roots), "leaves": len(leaves), "linear_chain": linear, "dot_print_rc": rc,
          "kernel_names_matched": bool(names), "loss_after_3_replays": float(loss),
          "allocator_events_recorded": len(history), "non_kernel_nodes": []}
for n in ids:
    if kind[n] != "kernel":
        report["non_kernel_nodes"].append({
            "node": describe(n), "pred": [describe(p) for p in pred[n]], "succ": [describe(s) for s in succ[n]],
            "blocks_overlapping_destination_in_allocation_order": owners.get(n, [])})
if args.tag and names:
    report["chain"] = [describe(n) for n in sorted(ids, key=lambda x: pos_in_chain.get(x, order[x]))]
with open(os.path.join(args.out, args.tag + ".json"), "w") as f:
    json.dump(report, f, indent=1)
print(json.dumps({k: v for k, v in report.items() if k not in ("non_kernel_nodes", "chain")}))
for r in report["non_kernel_nodes"]:
    print("  ", r["node"], "\n      pred:", r["pred"], "\n      succ:", r["succ"])
    for h in r["blocks_overlapping_destination_in_allocation_order"][-12:]:
        print("      block", h["action"], hex(h["addr"]), h["size"], "stream", h["stream"], h["where"])
