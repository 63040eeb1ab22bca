#!/usr/bin/env python
"""Headline benchmark: KGE triples scored / s on BASELINE.json configs[1]
(RotatE dim=200, 256 negatives per positive, ml-1m-small-shaped synthetic KG), per GPU and whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One *batch* = one pass of the scoring hot path over 2048 (user, positive item) groups, inputs resident in HBM:
    1. device negative sampling: 256 negatives per group over the item rows [1, n_items), rejecting the user's
       interacted items;
    2. grouped RotatE scorer over 2048 x (1 + 256) = 526 336 (h, r, t) triples;
    3. self-adversarial negative-weighting loss over the [2048, 257] scores
   -- all three in ONE launch (hw_kge_score_grouped_sampled_selfadv_chained: the one-block sum of batch i's 2048 group
   terms rides in block 0 of the next launch, the step's last ones are flushed by hw_selfadv_finish inside the timed
   region; --unchained-loss / --unfused-loss / --unfused-sampler run the pieces as the separate launches they equal bit
   for bit).  Consecutive batches go to alternating HIP streams (--streams, default 2; own id / score buffers per stream):
   a launch is one generation of workgroups, and the next batch fills its ramp-up and tail.
One *step* = `batches_per_step` such batches back to back (chosen so that the K timed steps hold >= 0.75 s of GPU
work whatever K is: a lone batch is 66 us).  value = triples scored per second over the whole job (all ranks).

--gpus N > 1: when not already under torch.distributed.run this process launches N ranks itself (one per GPU,
backend nccl = RCCL) BEFORE touching the GPU and relays rank 0's JSON line.  The headline stays configs[1] with the
32 MB of tables replicated and every rank scoring its own batches (the path shards by independent (h, r) groups:
weak scaling, no data-path collective -- DESIGN.md §5); the line also carries the `sharded` leg = configs[3]
(ComplEx d=256, Yelp2018 shape, user / entity tables row-sharded, RCCL all-to-all lookup + reverse all-to-all of the
row gradients + all-reduce of the replicated relation tables inside the timed step), which is what exercises xGMI.

Legs in the JSON line: `roofline` (the scorer kernel: HBM-resident variant as the roofline of record, the configs[1]
launch against the L2 gather ceiling, the entity-tail variants, `other_scorers` = the same launch shape for TransE d=64 / d=200,
DistMult d=200, ComplEx d=256 on >= 4 GB tables), `configs0` (TransE d=64 in the reference's layout), `evaluation` (the RotatE full-sort + top-10 step, product form vs gather kernel), `cpu_baseline` (oracle/kge_ref.c on the host
cores), `sharded` (configs[3]), `tucker` (configs[4], N = 1), `reference_layout` (configs[1] in the reference's batch
layout through the model class, N = 1), `pgpr` (configs[2]).
"""

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM, K_NEG, B_POS = 200, 256, 2048
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_GATHER_PEAK_GBS = 17800.0  # MI355X_MICROARCH.md 'Indexed rows': rows shared in an XCD's L2, 16.8-18.8 TB/s chip-wide
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix rate (MI355X_MICROARCH.md); a three-way split fp32 product costs 6 bf16 MFMAs
PLAIN_KERNEL = "kge_score_grouped_kernel<1, 64, 4, true, 4, false"   # RotatE d=200 instantiations as rocprofv3 names them
FUSED_KERNEL = "kge_score_grouped_kernel<1, 64, 4, true, 4, true, true"
TARGET_TIMED_SECONDS = 0.75


def algorithmic_bytes_per_triple(dim=DIM, k=K_NEG):
    """tail rows 2*d*4 + tail id 8 + score 4 + shared (h_re, h_im, theta rows + 2 ids)/(K+1)."""
    return 2 * dim * 4 + 8 + 4 + (3 * dim * 4 + 16) / (k + 1)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batches-per-step", type=int, default=0, help="0 = sized so the timed region is >= 0.75 s")
    ap.add_argument("--workload", choices=["replica", "sharded"], default="replica",
                    help="which leg is the headline value: configs[1] replicas (default) or configs[3] row-sharded ComplEx")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-variant", action="store_true")
    ap.add_argument("--no-pgpr", action="store_true")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--no-tucker", action="store_true")
    ap.add_argument("--no-cafe", action="store_true")
    ap.add_argument("--unfused-sampler", action="store_true", help="run hw_neg_sample as its own launch before the scorer")
    ap.add_argument("--unfused-loss", action="store_true", help="run hw_loss_self_adversarial as its own launches after the scorer")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the timed batches alternate over (each with its own id / score buffers and loss chain)")
    ap.add_argument("--unchained-loss", action="store_true",
                    help="finish every batch's loss with its own one-block launch instead of inside the next batch's launch")
    ap.add_argument("--pgpr-steps", type=int, default=0, help="0: enough searches for 0.5 s of timed work (>= 5)")
    ap.add_argument("--pgpr-users", type=int, default=256,
                    help="users per beam search: 64 -> 5.8 G paths/s, 256 -> 6.7 G (the first hops' fixed cost amortised), ~8 GB of work buffers")
    ap.add_argument("--pgpr-streams", type=int, default=3,
                    help="HIP streams the timed PGPR searches (independent user batches) are dealt over")
    ap.add_argument("--cafe-users", type=int, default=2048,
                    help="users per CAFE run_program call (hopwise's eval_batch_size default is 4 096)")
    ap.add_argument("--cafe-streams", type=int, default=2, help="HIP streams the timed CAFE calls are dealt over")
    ap.add_argument("--sharded-steps", type=int, default=10)
    ap.add_argument("--sharded-batch", type=int, default=8192, help="rec positives and kg triples per rank per step")
    ap.add_argument("--sharded-deadline", type=float, default=300.0,
                    help="N > 1: seconds the row-sharded leg may take before the line is printed without it")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks (before any GPU call in this process) and relay
    rank 0's JSON line."""
    import torch  # device_count() does not initialise the GPU

    backend = os.environ.get("HW_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s); RCCL needs one device per rank "
                 f"(HW_BENCH_BACKEND=gloo rehearses the multi-process control flow with ranks sharing a device)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)  # kept whatever the exit code: a timed-out sharded leg still leaves the replica numbers
    if p.returncode != 0 or line is None:
        sys.exit(f"bench.py: the {args.gpus}-rank run failed (exit code {p.returncode})")


class Dist:
    """The process group of this run (or a single rank)."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}; launch with --nproc-per-node {args.gpus}")
        import torch

        self.backend = None
        if self.world > 1:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # backend "nccl" is RCCL on ROCm; HW_BENCH_BACKEND=gloo exists only to rehearse the multi-process control
            # flow with several ranks sharing one GPU (RCCL refuses two ranks on one device)
            self.backend = os.environ.get("HW_BENCH_BACKEND", "nccl")
            n_dev = max(torch.cuda.device_count(), 1)
            if self.backend == "nccl" and n_dev < self.world:
                sys.exit(f"bench.py: {self.world} ranks but {n_dev} GPU(s): RCCL needs one device per rank")
            torch.cuda.set_device(self.local_rank % n_dev)
            dist.init_process_group(backend=self.backend)
            self.dist = dist
        self.dev = torch.device("cuda", self.local_rank % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(self.dev)
        self.torch = torch

    @property
    def rdev(self):
        return self.dev if self.backend == "nccl" else "cpu"

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, x, op="max"):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.rdev)
        self.dist.all_reduce(t, op={"max": self.dist.ReduceOp.MAX, "sum": self.dist.ReduceOp.SUM}[op])
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed(D, fn, steps, warmup):
    """W untimed + K timed calls of fn(i), bracketed by barrier + synchronize; max over ranks of the wall time."""
    for i in range(warmup):
        fn(i)
    D.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    D.barrier()
    return D.reduce(time.perf_counter() - t0, "max")


# ---------------------------------------------------------------------------------------------------------------------
# configs[1]: RotatE grouped scorer + in-kernel sampler + self-adversarial loss
# ---------------------------------------------------------------------------------------------------------------------
def leg_kge(D, args, out):
    import ctypes as C

    import numpy as np

    torch, dev, rank = D.torch, D.dev, D.rank
    from hopwise_amd import _lib, ops
    from hopwise_amd.data import synthetic

    kg = synthetic.make("ml-1m-small", seed=2024)
    g = torch.Generator(device="cpu").manual_seed(2024)

    def xavier(rows):
        return (torch.randn(rows, DIM, generator=g) * (2.0 / (rows + DIM)) ** 0.5).to(dev)

    user_re, user_im = xavier(kg.n_users), xavier(kg.n_users)
    ent_re, ent_im = xavier(kg.n_entities), xavier(kg.n_entities)
    theta = (torch.rand(kg.n_relations, DIM, generator=g) * (2 * np.pi)).to(dev)
    head, rel, tail = (user_re, user_im), (theta,), (ent_re, ent_im)
    # rec side: tails are items, i.e. entity rows [0, n_items) (what the model class passes for `side == "rec"`,
    # knowledge_graph_embedding_recommender.py _score; the reference draws rec negatives over items, sampler.py:226-227)
    desc = ops.make_desc("RotatE", head, rel, tail, margin=1.0, tail_rows=kg.n_items)

    rng = np.random.default_rng(2024 + rank)  # batches differ per rank
    pick = rng.integers(0, len(kg.inter_users), B_POS)
    users = torch.from_numpy(kg.inter_users[pick]).to(dev)
    pos = torch.from_numpy(kg.inter_items[pick]).to(dev)
    uptr, uidx = synthetic.used_csr(kg.inter_users, kg.inter_items, kg.n_users)
    uptr, uidx = torch.from_numpy(uptr).to(dev), torch.from_numpy(uidx).to(dev)
    lib = _lib.load()
    ws = torch.empty(int(lib.hw_loss_workspace_floats()) + B_POS, dtype=torch.float32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    seed = C.c_uint64(2024 + 7919 * rank)  # per-rank Philox key (streams keyed by (seed, rank, step))
    fused = not args.unfused_sampler
    fused_loss = fused and not args.unfused_loss
    chained = fused_loss and not args.unchained_loss

    # Lanes: batch i is launched on HIP stream i % n_lanes, each lane with its own id matrix, score matrix, loss slot and
    # loss chain.  One launch is exactly one generation of workgroups (2048 groups = 256 CUs x 8), so a lone stream leaves
    # the ramp-up, the tail and the gap between launches idle; the next batch on a second stream fills them (51 -> 40 us per
    # batch).  --streams 1 is the single-stream schedule.
    class Lane:
        pass

    def make_lane(torch_stream):
        ln = Lane()
        ln.torch_stream = torch_stream
        ln.stream = C.c_void_p(torch_stream.cuda_stream)
        ln.t_idx = torch.empty((B_POS, 1 + K_NEG), dtype=torch.int64, device=dev)  # one row of tail ids per (h, r) group
        ln.t_idx[:, 0] = pos
        neg_view = ln.t_idx[:, 1:]
        ln.scores = torch.empty((B_POS, 1 + K_NEG), dtype=torch.float32, device=dev)
        ln.loss = torch.empty(1, dtype=torch.float32, device=dev)
        ln.ws = torch.empty_like(ws)
        ln.group_ws = (torch.empty(B_POS, dtype=torch.float32, device=dev), torch.empty(B_POS, dtype=torch.float32, device=dev))
        ln.cur, ln.pending = 0, False
        t, sc = ln.t_idx, ln.scores
        ln.samp_head = (P(uptr), P(uidx), uptr.numel() - 1, P(users), B_POS, K_NEG, kg.n_items, None, None, None, 0, seed)
        ln.samp_tail = (P(neg_view), neg_view.stride(0), neg_view.stride(1), P(oob), ln.stream)
        ln.score_args = (C.byref(desc), P(users), None, kg.ui_relation, P(t), t.stride(0), t.stride(1), B_POS, 1 + K_NEG, P(sc),
                         sc.stride(0), sc.stride(1), P(oob), ln.stream)
        ln.loss_args = (P(sc), B_POS, 1 + K_NEG, C.c_float(1.0), P(ln.loss), None, P(ln.ws), ln.stream)
        # sampler + scorer in one launch ...
        ln.fused_args = (C.byref(desc), P(users), None, kg.ui_relation, P(t), t.stride(0), t.stride(1), B_POS, 1 + K_NEG,
                         P(uptr), P(uidx), uptr.numel() - 1, P(users), kg.n_items, seed)
        ln.fused_tail = (P(sc), sc.stride(0), sc.stride(1), P(oob), ln.stream)
        # ... and the loss inside the same launch (the block that scores a group reduces it):
        # hw_kge_score_grouped_sampled_selfadv = the separate calls, bit for bit (tests/test_gpu_sampler.py) ...
        ln.loss_tail = (P(sc), sc.stride(0), sc.stride(1), C.c_float(1.0), P(ln.loss), P(ln.group_ws[0]), None, P(oob), ln.stream)
        # ... and as a chain over the lane's batches (--unchained-loss: one finish launch per batch): batch i's 2048 group
        # terms are added by block 0 of the lane's next launch, the last one by hw_selfadv_finish inside the timed region, so
        # every batch's loss exists when the clock stops (equal to the unchained losses bit for bit, same test file)
        ln.chain_head = (P(sc), sc.stride(0), sc.stride(1), C.c_float(1.0))
        return ln

    n_lanes = max(1, args.streams)
    # lanes on streams that were probed to sit on different hardware queues (hopwise_amd/stream_pool.py: torch's pooled streams
    # share four queues, and two lanes that take turns on one queue are one lane)
    from hopwise_amd.stream_pool import concurrent_streams

    lane_streams = [torch.cuda.current_stream()] if n_lanes == 1 else concurrent_streams(n_lanes, dev)
    lanes = [make_lane(s) for s in lane_streams]
    solo = lanes[0] if n_lanes == 1 else make_lane(torch.cuda.current_stream())  # single-stream launches (calibration, later legs)
    torch.cuda.synchronize()
    if fused_loss and lib.hw_kge_score_grouped_sampled_selfadv(*solo.fused_args, C.c_uint64(0), *solo.loss_tail) != 0:
        fused_loss = chained = False

    def chained_batch(ln, i):
        prev = (P(ln.group_ws[ln.cur ^ 1]), B_POS, P(ln.loss)) if ln.pending else (None, 0, None)
        rc = lib.hw_kge_score_grouped_sampled_selfadv_chained(*ln.fused_args, C.c_uint64(i), *ln.chain_head,
                                                              P(ln.group_ws[ln.cur]), *prev, P(oob), ln.stream)
        ln.cur, ln.pending = ln.cur ^ 1, True
        return rc

    def chain_flush(which=None):
        for ln in (which or lanes):
            if ln.pending:
                if lib.hw_selfadv_finish(P(ln.group_ws[ln.cur ^ 1]), B_POS, P(ln.loss), ln.stream):
                    raise RuntimeError("hw_selfadv_finish failed")
                ln.pending = False

    events = []

    n_launched = [1 if fused_loss else 0]  # the validation launch above

    def batch(i, ev=None, ln=None):
        ln = ln or lanes[i % n_lanes]
        n_launched[0] += 1
        rc = 0
        if not fused:
            rc = lib.hw_neg_sample(*ln.samp_head, C.c_uint64(i), *ln.samp_tail)
        if ev is not None:
            ev[0].record(ln.torch_stream)
        if chained:
            rc |= chained_batch(ln, i)
        elif fused_loss:
            rc |= lib.hw_kge_score_grouped_sampled_selfadv(*ln.fused_args, C.c_uint64(i), *ln.loss_tail)
        elif fused:
            rc |= lib.hw_kge_score_grouped_sampled(*ln.fused_args, C.c_uint64(i), *ln.fused_tail)
        else:
            rc |= lib.hw_kge_score_grouped(*ln.score_args)
        if ev is not None:
            ev[1].record(ln.torch_stream)
        if not fused_loss:
            rc |= lib.hw_loss_self_adversarial(*ln.loss_args)
        if rc:
            raise RuntimeError(f"launch failed: {rc}")

    # what an event pair costs by itself (record -> record with nothing between): subtracted from the kernel's pairs
    empty = []
    for _ in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        empty.append((e0, e1))
    torch.cuda.synchronize()
    pair_overhead_ms = float(np.median([a.elapsed_time(b) for a, b in empty]))
    # size a step: time 64 batches, then enough batches per step for >= TARGET_TIMED_SECONDS over the K steps
    for i in range(16):
        batch(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(64):
        batch(16 + i)
    chain_flush()
    torch.cuda.synchronize()
    t_batch = (time.perf_counter() - t0) / 64
    inner = args.batches_per_step or max(1, math.ceil(TARGET_TIMED_SECONDS / (max(args.steps, 1) * t_batch)))
    inner = int(D.reduce(inner, "max"))
    EVERY = 16  # HIP events (on the launch stream) around every 16th scorer launch of the timed region

    def step(s):
        base = s * inner
        timing = s >= args.warmup
        for j in range(inner):
            ev = None
            if timing and j % EVERY == 0:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events.append(ev)
            batch(base + j, ev)
        chain_flush()  # the step's last loss, inside the timed region

    with torch.no_grad():
        dt = timed(D, step, args.steps, args.warmup)
    assert all(torch.isfinite(ln.loss).all().item() for ln in lanes)
    assert int(oob.item()) == 0 and ops.oob_count(dev) == 0
    # the launch by itself: the same kernel back to back on ONE stream, event pairs around every 4th launch (untimed pass)
    overlapped_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else None
    if n_lanes > 1:
        events.clear()
        for j in range(64 + 512):
            ev = None
            if j >= 64 and j % 4 == 0:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                events.append(ev)
            batch(10 ** 6 + j, ev, solo)
        chain_flush([solo])
        torch.cuda.synchronize()
    t_idx, scores, loss = solo.t_idx, solo.scores, solo.loss
    _log_launches("timed_region_and_single_stream_pass",
                  FUSED_KERNEL if fused_loss else "kge_score_grouped_kernel<1, 64, 4, true, 4, true, false" if fused else PLAIN_KERNEL,
                  n_launched[0])

    triples_per_batch = B_POS * (1 + K_NEG)
    triples_per_step = triples_per_batch * inner
    # launch duration from the event pairs.  A pair around nothing measures `pair_overhead_ms` (reported); the pair is NOT
    # corrected by it -- record-to-record time is an upper bound of the kernel's duration, so `achieved` is conservative
    # (rocprofv3's kernel trace of the same command reads ~6 % shorter, profiles/r2/bench_kernel_stats_*.csv)
    kern_ms = kern_ms_raw = float(np.mean([a.elapsed_time(b) for a, b in events]))
    abytes = algorithmic_bytes_per_triple() * triples_per_batch
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))
        except Exception:
            traffic = {}
    out.update({
        "metric": "KGE triples scored per second (RotatE d=200, 256 negatives; sampler + scorer + self-adversarial loss)",
        "value": D.world * triples_per_step * args.steps / dt,
        "unit": "triples/s",
        "n_gpus": D.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: RotatE dim=200, 256 negatives, ml-1m-small-shaped KG; grouped scorer B=2048 x 257 per "
                        f"batch, {inner} batches per step; rec side: negative tails drawn over the {kg.n_items} ITEM rows "
                        f"(entity rows [1, n_items), {2 * kg.n_items * DIM * 4 / 1e6:.1f} MB of tail rows, L2-resident) -- "
                        "see roofline.entity_tails for randint(1, n_entities)",
            "n_users": kg.n_users, "n_items": kg.n_items, "n_entities": kg.n_entities, "n_relations": kg.n_relations,
            "triples_per_batch": triples_per_batch, "batches_per_step": inner, "triples_per_step": triples_per_step,
            "us_per_batch": dt / args.steps / inner * 1e6,
            "streams": n_lanes,
            "launches_per_batch": ("1 (sampler + scorer + loss; the 2048-term sum of a batch rides in the next batch's launch, "
                                   "one hw_selfadv_finish per step)" if chained else
                                   "2 (sampler + scorer + per-group loss, one-block finish)" if fused_loss else
                                   "3 (sampler + scorer, loss, finish)" if fused else "4"),
            "parallelism": f"replicas x{D.world} (tables replicated, batches per rank, no data-path collective)",
        },
    })
    l2 = {
        "bound": "l2", "kernel": "kge_score_grouped_kernel<RotatE" + (", SAMPLE, LOSS>" if fused_loss else ", SAMPLE>" if fused else ">"),
        "achieved": abytes / (kern_ms * 1e-3) / 1e9, "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s",
        "frac": abytes / (kern_ms * 1e-3) / 1e9 / L2_GATHER_PEAK_GBS,
        "traffic": traffic.get("kge_score_grouped_bytes_per_launch"), "kernel_ms": kern_ms,
        "event_pair_ms_raw": kern_ms_raw, "empty_event_pair_ms": pair_overhead_ms,
        "algorithmic_bytes_per_launch": abytes, "events": len(events),
        "scorer_only_triples_per_s": triples_per_batch / (kern_ms * 1e-3),
        "note": "the launch of the timed region, measured back to back on ONE stream (untimed pass after the timed region when "
                "the timed batches alternate over several streams): its 4.8 MB of tail rows are served by the XCDs' L2s, so "
                "the ceiling is the guide's L2 gather rate (16.8-18.8 TB/s chip-wide), not HBM; the kernel is load-issue bound there",
    }
    if n_lanes > 1:
        us = dt / args.steps / inner * 1e6
        l2["overlapped"] = {
            "streams": n_lanes, "us_per_batch": us, "launch_ms_under_overlap": overlapped_ms,
            "aggregate_algorithmic_GBps": abytes / (us * 1e-6) / 1e9,
            "note": "timed region: consecutive batches on alternating HIP streams.  One launch is exactly one generation of "
                    "workgroups, so a lone stream idles during ramp-up, tail and the launch gap; a second stream fills them.  "
                    "The aggregate rate is bytes per batch / wall time per batch, not a per-launch figure (a launch takes "
                    "launch_ms_under_overlap while it shares the chip)"}
    out["roofline"] = {"configs1_launch": l2}
    ctx = dict(kg=kg, head=head, rel=rel, tail=tail, users=users, t_idx=t_idx, desc=desc, abytes=abytes,
               uptr_np=uptr.cpu().numpy(), uidx_np=uidx.cpu().numpy(),
               launch_scores=(scores if fused else None), launch_loss=(loss if fused_loss else None),
               triples_per_batch=triples_per_batch, traffic=traffic, tabs=(user_re, user_im, theta, ent_re, ent_im))
    return ctx


LAUNCH_LOG = []  # [{"tag", "kernel", "launches"}] in launch order; written to $HW_BENCH_LAUNCH_LOG when set


def _log_launches(tag, kernel, n):
    LAUNCH_LOG.append({"tag": tag, "kernel": kernel, "launches": int(n)})


def _time_kernel(torch, fn, min_seconds=0.3, reps0=20, tag=None, kernel=None):
    """Average launch duration (ms) of fn() from HIP events on the launch stream, over >= min_seconds of launches.
    With ``tag`` the launches are recorded in LAUNCH_LOG: the variants of one kernel run under one name, and the committed
    profile summary (tools/summarize_profile.py) splits the kernel trace by this log -- one row per variant."""
    for _ in range(3):
        fn()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record()
    for _ in range(reps0):
        fn()
    eb.record()
    torch.cuda.synchronize()
    ms = ea.elapsed_time(eb) / reps0
    reps = max(reps0, int(min_seconds * 1e3 / max(ms, 1e-3)))
    ea.record()
    for _ in range(reps):
        fn()
    eb.record()
    torch.cuda.synchronize()
    if tag:
        _log_launches(tag, kernel, 3 + reps0 + reps)
    return ea.elapsed_time(eb) / reps, reps


def leg_roofline_variants(D, args, out, ctx):
    """Same kernel, same batch shape, other tail spaces: (a) the HBM-resident table = the roofline of record;
    (b) SURVEY §8(d) config 2 (ii): neg_idx ~ randint(1, n_entities) over the 22 MB entity tables, plain and
    XCD-partitioned kernels."""
    torch, dev = D.torch, D.dev
    from hopwise_amd import ops

    kg, users, abytes, tpb = ctx["kg"], ctx["users"], ctx["abytes"], ctx["triples_per_batch"]
    gen = torch.Generator(device=dev).manual_seed(2024)
    # (b) entity tails
    edesc = ops.make_desc("RotatE", ctx["head"], ctx["rel"], ctx["tail"], margin=1.0)
    eidx = torch.randint(1, kg.n_entities, (B_POS, 1 + K_NEG), device=dev, dtype=torch.int64, generator=gen)
    ent = {"tail_space": f"randint(1, {kg.n_entities}) over both entity tables ({2 * kg.n_entities * DIM * 4 / 1e6:.1f} MB: "
                         "beyond one XCD's 4 MiB L2, inside the Infinity Cache)"}
    prev = os.environ.get("HW_GROUPED_VARIANT")
    for name, var in (("plain", "0"), ("xcd_partitioned", "1")):
        os.environ["HW_GROUPED_VARIANT"] = var
        ms, reps = _time_kernel(torch, lambda: ops.kge_score_grouped_raw(edesc, users, None, kg.ui_relation, eidx),
                                tag=f"entity_tails_{name}", kernel="kge_score_grouped_xcd_kernel<1," if var == "1" else PLAIN_KERNEL)
        ent[name] = {"kernel_ms": ms, "launches": reps, "achieved": abytes / (ms * 1e-3) / 1e9,
                     "triples_per_s": tpb / (ms * 1e-3),
                     "traffic": ctx["traffic"].get(f"kge_score_grouped_entity_tails_{'xcd' if var == '1' else 'plain'}_bytes_per_launch")}
    if prev is None:
        os.environ.pop("HW_GROUPED_VARIANT", None)
    else:
        os.environ["HW_GROUPED_VARIANT"] = prev
    ent["note"] = ("GB/s are algorithmic bytes / kernel time; the launcher picks the XCD-partitioned kernel by itself at "
                   "this footprint (block b scores only tails with id % 8 == b % 8, so each XCD's L2 holds 1/8 of the table)")
    out["roofline"]["entity_tails"] = ent
    if args.no_hbm_variant:
        out["roofline"].update(out["roofline"]["configs1_launch"])
        return
    # (a) an entity table far beyond L2 + Infinity Cache (2.6 M rows x 2 tables x 800 B = 4.2 GB), negatives drawn over
    # all of it: every tail row is an HBM miss.  This is the number that may be compared with the 8 TB/s peak.
    nbig = 2_621_440
    big_re = torch.randn(nbig, DIM, device=dev) * 0.01
    big_im = torch.randn(nbig, DIM, device=dev) * 0.01
    bdesc = ops.make_desc("RotatE", ctx["head"], ctx["rel"], (big_re, big_im), margin=1.0)
    bidx = torch.randint(1, nbig, (B_POS, 1 + K_NEG), device=dev, dtype=torch.int64, generator=gen)
    ms, reps = _time_kernel(torch, lambda: ops.kge_score_grouped_raw(bdesc, users, None, kg.ui_relation, bidx),
                            tag="hbm_plain", kernel=PLAIN_KERNEL)
    ach = abytes / (ms * 1e-3) / 1e9
    plain = {"kernel": "kge_score_grouped_kernel<RotatE> (pre-drawn ids, no loss)", "kernel_ms": ms, "launches": reps,
             "achieved": ach, "frac": ach / HBM_PEAK_GBS, "triples_per_s": tpb / (ms * 1e-3),
             "traffic": ctx["traffic"].get("kge_score_grouped_hbm_variant_bytes_per_launch")}
    # the instantiation of the TIMED REGION (sampler + scorer + self-adversarial loss in one launch) on the same 4.2 GB table:
    # the used sets keep the real interaction counts per user, their ids spread over the big id space (sorted per user: the
    # fused sampler searches the slice beyond the 65 536-id LDS bitmap)
    import numpy as np

    counts = np.diff(ctx["uptr_np"])
    rng = np.random.default_rng(7)
    big_used = rng.integers(1, nbig, int(counts.sum()))
    seg = np.repeat(np.arange(len(counts)), counts)
    big_used = big_used[np.lexsort((big_used, seg))]
    bptr = torch.from_numpy(ctx["uptr_np"]).to(dev)
    bused = torch.from_numpy(big_used).to(dev)
    t_big = bidx.clone()
    state = {"i": 0}
    # exactly the launch of the timed region: the chained form (batch i's 2048 group terms are added by block 0 of batch
    # i + 1, no finish launch per batch), preallocated scores / loss
    chain = ops.SelfAdvChain(B_POS, dev)
    f_out = torch.empty((B_POS, 1 + K_NEG), dtype=torch.float32, device=dev)
    f_loss = torch.empty(1, dtype=torch.float32, device=dev)

    def fused_launch():
        state["i"] += 1
        chain.launch(bdesc, users, None, kg.ui_relation, t_big, bptr, bused, users, nbig, 2024, state["i"], alpha=1.0,
                     out=f_out, loss=f_loss)

    fused = None
    try:
        fms, freps = _time_kernel(torch, fused_launch, tag="hbm_fused", kernel=FUSED_KERNEL)
        chain.flush()
        torch.cuda.synchronize()
        assert torch.isfinite(f_loss).all().item()
        fach = abytes / (fms * 1e-3) / 1e9
        fused = {"kernel": "kge_score_grouped_kernel<RotatE, SAMPLE, LOSS> (the timed region's instantiation)",
                 "kernel_ms": fms, "launches": freps, "achieved": fach, "frac": fach / HBM_PEAK_GBS,
                 "triples_per_s": tpb / (fms * 1e-3),
                 "traffic": ctx["traffic"].get("kge_score_grouped_hbm_fused_bytes_per_launch"),
                 "note": "negatives drawn in the launch (Philox + rejection against the user's used slice, binary search: the id "
                         "space is beyond the LDS bitmap), 257 scores per group reduced to the self-adversarial loss term in the "
                         "same block, chained like the timed region (hw_kge_score_grouped_sampled_selfadv_chained)"}
    except Exception as e:  # noqa: BLE001
        fused = {"error": f"{type(e).__name__}: {e}"[:300]}
    head = fused if fused and "achieved" in fused else plain
    rec = {
        "bound": "hbm", "kernel": head["kernel"], "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": head["achieved"] / HBM_PEAK_GBS, "traffic": head["traffic"],
        "kernel_ms": head["kernel_ms"], "launches": head["launches"], "algorithmic_bytes_per_launch": abytes,
        "triples_per_s": head["triples_per_s"],
        "fused_instantiation": fused, "plain_scorer": plain,
        "workload": f"the configs[1] kernel and batch shape (2048 x 257 triples per launch) on a {2 * nbig * DIM * 4 / 1e9:.1f} GB "
                    f"entity table ({nbig} rows), tails ~ randint(1, {nbig}): every tail row is an HBM miss.  configs[1]'s "
                    "own tables (32 MB) never leave L2 / Infinity Cache, so its launch is reported under configs1_launch "
                    "against the L2 gather ceiling instead.  The roofline of record is the instantiation the timed region "
                    "launches (fused_instantiation); plain_scorer is the same kernel without sampler and loss",
        "gather_ceiling_note": "the guide's measured random-row gather from HBM is 5.5-5.8 TB/s (1.1-2.3 KB rows)",
        "traffic_source": ctx["traffic"].get("source"),
        "traffic_note": (f"traffic / algorithmic = {head['traffic'] / abytes:.3f} (PMC pass: {ctx['traffic'].get('source')}): "
                         if head["traffic"] else "no PMC pass on file: ") +
                        "an 800-byte row (d = 200 floats) always occupies 7 lines of 128 B (6.25 needed) whatever its "
                        "alignment, and RotatE reads two such rows (re, im tables) per tail: 14 lines for 1 600 B = 1.12x, plus the "
                        "id matrix and the scores.  Not re-reads, and not what bounds the launch: with re | im interleaved in one "
                        "1 600-byte row (13 lines; row-strided descriptor, measured) the same launch takes the same time "
                        "(DESIGN.md \u00a78)",
    }
    # the same launches alternating over two streams (what the timed region of the headline does): aggregate rate only --
    # the roofline of record stays the lone launch above
    from hopwise_amd.stream_pool import concurrent_streams

    st = concurrent_streams(2, dev)  # two streams on two hardware queues
    torch.cuda.synchronize()

    def pair(n):
        for i in range(n):
            with torch.cuda.stream(st[i & 1]):
                ops.kge_score_grouped_raw(bdesc, users, None, kg.ui_relation, bidx)

    pair(8)
    torch.cuda.synchronize()
    n_pair = max(16, int(0.3 / (ms * 1e-3)))
    t0 = time.perf_counter()
    pair(n_pair)
    torch.cuda.synchronize()
    _log_launches("hbm_plain_two_streams", PLAIN_KERNEL, 8 + n_pair)
    per = (time.perf_counter() - t0) / n_pair
    rec["two_streams_aggregate"] = {"us_per_launch": per * 1e6, "GBps": abytes / per / 1e9, "frac_of_peak": abytes / per / 1e9 / HBM_PEAK_GBS,
                                    "launches": n_pair, "note": "wall time per launch with two launches sharing the chip; "
                                    "not a per-kernel duration"}
    # "also report vs a measured device-to-device copy on the box" -- one big table copied onto the other
    # (2.1 GB read + 2.1 GB written per copy), timed like the kernels
    cms, creps = _time_kernel(torch, lambda: big_im.copy_(big_re), min_seconds=0.15, reps0=5)
    copy_gbs = 2 * big_re.numel() * 4 / (cms * 1e-3) / 1e9
    rec["measured_copy"] = {"GBps": copy_gbs, "ms": cms, "copies": creps, "bytes_moved_per_copy": 2 * big_re.numel() * 4,
                            "achieved_over_copy": ach / copy_gbs,
                            "note": "torch copy_ of one 2.1 GB table onto the other (read + write counted): what a plain "
                                    "streaming kernel moves on this box, next to the 8 TB/s vendor peak"}
    cfg1 = out["roofline"].pop("configs1_launch")
    ent = out["roofline"].pop("entity_tails")
    out["roofline"] = {**rec, "configs1_launch": cfg1, "entity_tails": ent}
    del big_re, big_im, bidx, t_big, bused
    torch.cuda.empty_cache()


SCORER_MODEL_ID = {"TransE": 0, "RotatE": 1, "ComplEx": 2, "DistMult": 3}


def scorer_algorithmic_bytes_per_triple(model, dim, k=K_NEG):
    """SURVEY.md \u00a78(d): tail rows NT*d*4 + tail id 8 + score 4 + the group's shared (NH + NR) rows and 2 ids over its K + 1 triples."""
    nh, nr, nt = {"TransE": (1, 1, 1), "DistMult": (1, 1, 1), "RotatE": (2, 1, 2), "ComplEx": (2, 2, 2)}[model]
    return nt * dim * 4 + 8 + 4 + ((nh + nr) * dim * 4 + 16) / (k + 1)


def leg_scorer_rooflines(D, args, traffic):
    """HBM rooflines of the other three named scorers (north star: TransE / RotatE / ComplEx / DistMult forward) and of
    configs[0]'s width: the grouped launch of the headline's shape (2048 groups x (1 + 256) tails, pre-drawn ids) on tail
    tables of >= 4 GB, tails ~ randint over all rows -- every tail row an HBM miss, as for the RotatE roofline of record."""
    torch, dev = D.torch, D.dev
    from hopwise_amd import ops

    gen = torch.Generator(device=dev).manual_seed(77)
    res = {"shape": f"{B_POS} groups x (1 + {K_NEG}) tails per launch, ids pre-drawn, tail tables >= 4 GB",
           "traffic_source": traffic.get("source")}
    tpb = B_POS * (1 + K_NEG)
    n_head, n_rel = 6041, 13
    for tag, model, dim in (("transe_d64", "TransE", 64), ("transe_d200", "TransE", 200), ("distmult_d200", "DistMult", 200),
                            ("complex_d256", "ComplEx", 256)):
        nh, nr, nt = {"TransE": (1, 1, 1), "DistMult": (1, 1, 1), "ComplEx": (2, 2, 2)}[model]
        rows = int(math.ceil(4.2e9 / (nt * dim * 4) / 65536) * 65536)
        tail = tuple(torch.randn(rows, dim, device=dev) * 0.01 for _ in range(nt))
        head = tuple(torch.randn(n_head, dim, device=dev) * 0.01 for _ in range(nh))
        rel = tuple(torch.randn(n_rel, dim, device=dev) * 0.01 for _ in range(nr))
        desc = ops.make_desc(model, head, rel, tail, margin=1.0)
        users = torch.randint(1, n_head, (B_POS,), device=dev, dtype=torch.int64, generator=gen)
        idx = torch.randint(1, rows, (B_POS, 1 + K_NEG), device=dev, dtype=torch.int64, generator=gen)
        shape_g = 16 if dim // 4 <= 16 else (32 if dim // 4 <= 32 else 64)
        kname = f"kge_score_grouped_kernel<{SCORER_MODEL_ID[model]}, {shape_g}, 4, true, 4, false"
        ms, reps = _time_kernel(torch, lambda: ops.kge_score_grouped_raw(desc, users, None, n_rel - 1, idx),
                                tag=f"scorer_hbm_{tag}", kernel=kname)
        abytes = scorer_algorithmic_bytes_per_triple(model, dim) * tpb
        ach = abytes / (ms * 1e-3) / 1e9
        tr = traffic.get(f"scorer_hbm_{tag}_bytes_per_launch")
        res[tag] = {"bound": "hbm", "kernel": f"kge_score_grouped_kernel<{model}> d={dim}", "kernel_ms": ms, "launches": reps,
                    "algorithmic_bytes_per_triple": scorer_algorithmic_bytes_per_triple(model, dim),
                    "algorithmic_bytes_per_launch": abytes, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_over_algorithmic": (tr / abytes if tr else None),
                    "triples_per_s": tpb / (ms * 1e-3), "table_GB": nt * rows * dim * 4 / 1e9, "tail_rows": rows}
        del tail, head, rel, desc, idx
        torch.cuda.empty_cache()
    res["note"] = ("a row of d floats occupies ceil-ish(4 d / 128) + 1 lines of 128 B at a random 16-byte alignment (d = 64: 256 B = "
                   "2-3 lines; d = 200: 800 B = 7; d = 256: 1 KB = 8-9), so traffic / algorithmic is largest for the short rows; "
                   "the launch is bound by row requests in flight (DESIGN.md \u00a74), which is why the short-row scorers sit "
                   "furthest below the byte peak")
    return res


def leg_configs0(D, args):
    """BASELINE.json configs[0]: TransE d=64 on the ml-1m-small shape in the layout hopwise's KnowledgeBasedDataLoader yields
    with the reference's defaults (train_batch_size 2048, one negative per positive): 2048 rec (pos, neg) pairs + 2048 kg pairs
    = 8 192 scored triples per calculate_loss, through hopwise_amd.model.TransE -- forward, the trainer's step, its graph replay."""
    import numpy as np

    torch, dev = D.torch, D.dev
    import hopwise_amd.model as M
    from hopwise_amd.driver import GraphedTrainStep
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, make_config

    nU, nI, nE, nR, B, d = 6041, 2985, 13805, 13, 2048, 64
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    m = M.TransE(make_config(dev, d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), neg_item_id=rng.integers(1, nI, B),
        head_id=rng.integers(1, nE, B), relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B),
        neg_tail_id=rng.integers(1, nE, B)).items()}
    res = {"metric": "configs[0]: TransE d=64 calculate_loss in the reference's batch layout (8 192 triples per step), model-class path",
           "triples_per_step": 4 * B}

    def fwd():
        with torch.no_grad():
            return m.calculate_loss(it)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.calculate_loss(it)
        loss.backward()
        opt.step()
        return loss

    def wall(fn, reps):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    for name, fn in (("forward", fwd), ("forward_backward_adam", step)):
        dt = wall(fn, 200)
        res[name] = {"ms": dt * 1e3, "triples_per_s": 4 * B / dt}
    gstep = GraphedTrainStep(m, opt, it)
    dt = wall(lambda: gstep(it), 300)
    res["forward_backward_adam_graph_replay"] = {"ms": dt * 1e3, "triples_per_s": 4 * B / dt}
    # evaluation of the same model: every user of a 2 048-user batch against the 2 985 items, history mask, top-10
    users = torch.from_numpy(rng.integers(1, nU, 2048)).to(dev)
    hist = (torch.from_numpy(rng.integers(0, 2048, 100_000)).to(dev), torch.from_numpy(rng.integers(1, nI, 100_000)).to(dev))
    with torch.no_grad():
        dt = wall(lambda: m.full_sort_topk({"user_id": users}, 10, hist), 100)
    res["evaluation_top10_2048_users"] = {"ms": dt * 1e3, "scores_per_s": 2048 * nI / dt}
    res["note"] = ("configs[0] is the reference's CPU-runnable plumbing case; BASELINE.md's CPU probe of the reference (8 vCPU) is "
                   "of the order of 0.5 M triples/s forward")
    del gstep, m, opt
    torch.cuda.empty_cache()
    return res


def leg_evaluation(D, args):
    """The evaluation step a hopwise trainer runs per batch (trainer.py:716-735 + collector.py:173-183): every user of a 2 048-user
    batch against every item, history mask, top-10 -- RotatE d=200 (the north-star scorer) at ml-1m-small's 2 985 items and at
    Yelp2018's 45 539, as one product on the MFMA GEMM with the exact re-rank (the default) against the fused gather kernel;
    the public full_sort_predict (the [B, N] matrix hopwise's own collector ranks) beside it."""
    import numpy as np

    torch, dev = D.torch, D.dev
    import hopwise_amd.model as M
    from tests.helpers import FakeDataset, make_config

    def wall(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    res = {"metric": "RotatE d=200 evaluation step: 2 048 users x all items -> history mask -> top-10 (ms per step)"}
    rng = np.random.default_rng(0)
    B = 2048
    for tag, (nU, nI, nE, nR) in (("ml1m_2985_items", (6041, 2985, 13805, 13)), ("yelp2018_45539_items", (45920, 45539, 90962, 44))):
        torch.manual_seed(0)
        m = M.RotatE(make_config(dev, DIM, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev).eval()
        users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
        hist = (torch.from_numpy(rng.integers(0, B, 50 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 50 * B)).to(dev))
        inter = {"user_id": users}
        r = {"scores_per_step": B * nI}
        with torch.no_grad():
            gv, gi = m.full_sort_topk(inter, 10, hist)
            r["topk_product_form_ms"] = wall(lambda: m.full_sort_topk(inter, 10, hist), 50)
            r["full_sort_predict_product_form_ms"] = wall(lambda: m.full_sort_predict(inter), 50)
            m.gemm_full_sort = False
            kv, ki = m.full_sort_topk(inter, 10, hist)
            r["topk_gather_kernel_ms"] = wall(lambda: m.full_sort_topk(inter, 10, hist), 20)
            r["full_sort_predict_gather_kernel_ms"] = wall(lambda: m.full_sort_predict(inter), 20)
        # parity in the line: the product form's top-10 against the fused gather kernel's (positions may swap only where two
        # exact scores are within an ulp of each other: the two kernels sum a row in different orders)
        same = (gi == ki)
        gap = (kv[:, :-1] - kv[:, 1:]).abs() > 4e-7 * kv[:, :-1].abs() + 1e-7
        decided = torch.ones_like(same)
        decided[:, :-1] &= gap
        decided[:, 1:] &= gap
        r["parity"] = {"columns_equal": float(same.float().mean()), "columns_equal_where_scores_differ": bool(same[decided].all()),
                       "max_rel_value_diff": float(((gv - kv).abs() / kv.abs().clamp_min(1e-6)).max())}
        if not r["parity"]["columns_equal_where_scores_differ"] or r["parity"]["max_rel_value_diff"] > 1e-5:
            raise AssertionError(f"evaluation parity ({tag}): {r['parity']}")
        r["scores_per_s_product_form"] = B * nI / (r["topk_product_form_ms"] * 1e-3)
        res[tag] = r
        del m
        torch.cuda.empty_cache()
    res["note"] = ("product form: hw_kge_query_rows -> hw_gemm_f32 (bias -||t||^2/2) -> mask -> top-16 -> exact "
                   "re-score of the candidates by the gather scorer -> hw_topk_rerank (proof on the device, predicated gather "
                   "fallback); full_sort_predict: the plain product + hw_kge_dist_finish")
    return res


def leg_cpu_baseline(D, args, out, ctx):
    import numpy as np

    from hopwise_amd import ops
    from oracle import cref

    kg = ctx["kg"]
    tabs = [t.cpu().numpy() for t in ctx["tabs"]]
    h_np, t_np = ctx["users"].cpu().numpy(), np.ascontiguousarray(ctx["t_idx"].cpu().numpy())
    outb = np.empty(t_np.shape, np.float32)
    cref.score_grouped("RotatE", tabs[0:2], tabs[2:3], tabs[3:5], h_np, None, kg.ui_relation, t_np, 1.0, 0.0, outb)
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.cpu_seconds:
        cref.score_grouped("RotatE", tabs[0:2], tabs[2:3], tabs[3:5], h_np, None, kg.ui_relation, t_np, 1.0, 0.0, outb)
        reps += 1
    cdt = time.perf_counter() - t0
    out["cpu_baseline"] = {
        "value": reps * ctx["triples_per_batch"] / cdt, "unit": "triples/s", "cores": cref.threads(), "kind": "port",
        "sample": f"{reps} passes of the same 2048 x 257 grouped RotatE scoring batch (scorer only) through "
                  f"oracle/kge_ref.c (OpenMP) in {cdt:.1f} s; host has {os.cpu_count()} logical CPUs",
    }
    # full-size parity of the headline path, outside the timed region: the LAST launch of the single-stream pass left its
    # sampled ids, its scores and its loss behind (sampler + scorer + loss in one launch); the C restatement scored the same
    # 2048 x 257 ids above.  Reported in the line, and a failed check fails the bench.
    gpu_scores = ops.kge_score_grouped_raw(ctx["desc"], ctx["users"], None, kg.ui_relation, ctx["t_idx"]).cpu().numpy()
    np.testing.assert_allclose(gpu_scores, outb, rtol=1e-4, atol=1e-5)  # same inputs, same answers
    rel = np.abs(gpu_scores - outb) / np.maximum(np.abs(outb), 1e-3)
    parity = {"triples": int(outb.size), "max_rel_score_err": float(rel.max()), "max_abs_score_err": float(np.abs(gpu_scores - outb).max()),
              "rule": "scores of the full 2048 x 257 batch against oracle/kge_ref.c at 1e-4 relative (+1e-5 absolute)"}
    if ctx.get("launch_scores") is not None:  # what the one-launch form itself wrote
        ls = ctx["launch_scores"].cpu().numpy()
        parity["one_launch_scores_equal_plain_scorer"] = bool(np.array_equal(ls, gpu_scores))
        assert parity["one_launch_scores_equal_plain_scorer"], "sampler + scorer launch: scores differ from the plain scorer on its own ids"
    if ctx.get("launch_loss") is not None:
        from oracle import kge as okge

        want = float(okge.self_adversarial_loss(outb, 1.0))
        got = float(ctx["launch_loss"].item())
        parity["loss"] = {"device": got, "oracle_float64": want, "rel_err": abs(got - want) / abs(want)}
        assert parity["loss"]["rel_err"] < 1e-4, parity["loss"]
    # the sampled negatives: item ids, none of them in the key's used set (sampler.py:140-183's contract)
    uptr, uidx = ctx["uptr_np"], ctx["uidx_np"]
    neg = t_np[:, 1:]
    in_range = bool(((neg >= 1) & (neg < kg.n_items)).all())
    hits = 0
    for gi, u in enumerate(h_np):
        hits += int(np.isin(neg[gi], uidx[uptr[u]:uptr[u + 1]]).sum())
    parity["negatives"] = {"drawn": int(neg.size), "in_item_range": in_range, "in_used_set": hits,
                           "distinct_ids": int(np.unique(neg).size)}
    assert in_range and hits == 0, parity["negatives"]
    out["parity"] = parity


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: ComplEx d=256, Yelp2018 shape, user / entity tables row-sharded over the ranks
# ---------------------------------------------------------------------------------------------------------------------
def leg_sharded(D, args):
    import numpy as np

    torch, dev = D.torch, D.dev
    from hopwise_amd import parallel
    from hopwise_amd.data import synthetic
    from hopwise_amd.model import ComplEx
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, make_config

    nU, nI, nE, nR, _, _ = synthetic.SHAPES["yelp2018"]
    Dm, B, P = 256, args.sharded_batch, D.world
    torch.manual_seed(0)  # same full tables on every rank before sharding
    model = ComplEx(make_config(dev, Dm, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
    sharded = parallel.RowShardedKGE(model).to(dev)
    opt = FusedAdam(sharded.parameters(), lr=1e-3, grad_sinks="caller_syncs")  # sync_grads() all-reduces p.grad itself
    rng = np.random.default_rng(100 + D.rank)

    def batch():
        return {k: torch.from_numpy(rng.integers(1, hi, B)).to(dev) for k, hi in (
            ("user_id", nU), ("item_id", nI), ("neg_item_id", nI), ("head_id", nE), ("relation_id", nR - 1),
            ("tail_id", nE), ("neg_tail_id", nE))}

    batches = [batch() for _ in range(4)]
    last = {}

    def step(i):
        opt.zero_grad(set_to_none=True)
        loss = sharded.calculate_loss(batches[i % 4])
        loss.backward()
        sharded.sync_grads()
        opt.step()
        last["loss"] = loss

    # full-size parity, outside the timed region: on the same batch the sharded model's loss is the plain model's
    # (the exchange and the compact tables change where rows live, not what is computed)
    with torch.no_grad():
        l_sh = float(sharded.calculate_loss(batches[0]))
    parity = None
    if P == 1:
        torch.manual_seed(0)
        plain = ComplEx(make_config(dev, Dm, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
        with torch.no_grad():
            l_pl = float(plain.calculate_loss(batches[0]))
        if abs(l_sh - l_pl) > 1e-6 * abs(l_pl):
            raise AssertionError(f"sharded loss {l_sh} != plain loss {l_pl}")
        parity = {"sharded_loss": l_sh, "plain_loss": l_pl, "rule": "same batch, same tables: equal to 1e-6 relative"}
        del plain
    n_steps, n_warm = (args.steps, args.warmup) if args.workload == "sharded" else (args.sharded_steps, 3)
    dt = timed(D, step, n_steps, n_warm)
    sharded.check_overflow()
    loss = float(last["loss"].detach())
    assert math.isfinite(loss)
    # bytes one rank puts on the fabric per step: fixed-size all-to-alls, (P - 1) / P of every buffer is remote
    cap_e = parallel.capacity(5 * B, nE, P, sharded.capacity_factor)
    cap_u = parallel.capacity(B, nU, P, sharded.capacity_factor)
    remote = (P - 1) / P
    ids_bytes = remote * P * (2 + cap_e + cap_u) * 4  # one [count, local rows...] int32 message per peer and lookup
    row_bytes = remote * P * (cap_e + cap_u) * 2 * Dm * 4  # 2 tables (re, im) per lookup
    xbytes = ids_bytes + 2 * row_bytes  # rows forward + row gradients backward
    b0 = batches[0]
    ent_u = torch.unique(torch.cat([b0[k] for k in ("item_id", "neg_item_id", "head_id", "tail_id", "neg_tail_id")])).numel()
    usr_u = torch.unique(b0["user_id"]).numel()
    res = {
        "metric": "ComplEx d=256 training triples per second, user/entity tables row-sharded owner = id % P (Yelp2018 shape): "
                  "calculate_loss -> backward -> all-reduce of the relation tables -> Adam on the local shards",
        "value": P * 4 * B * n_steps / dt, "unit": "triples/s", "per_gpu": 4 * B * n_steps / dt,
        "n_gpus": P, "steps": n_steps, "warmup": n_warm, "ms_per_step": dt / n_steps * 1e3,
        "batch_per_rank": {"rec": B, "kg": B, "triples": 4 * B}, "loss": loss, "parity": parity,
        "exchange": {
            "capacity_rows_per_owner": {"entity": cap_e, "user": cap_u, "factor": sharded.capacity_factor},
            "distinct_rows_per_step_per_rank": {"entity": ent_u, "user": usr_u},
            "bytes_per_step_per_rank": xbytes,
            "xgmi_GBps_per_gpu_over_step": xbytes / (dt / n_steps) / 1e9,
            "collectives_per_step": "6 all-to-alls (per lookup -- entity ids, user ids -- one int32 [count, rows] message, one "
                                    "float message with both tables' rows forward, one with their gradients backward) + "
                                    "ONE all-reduce of the flattened replicated (relation) gradients + one 4-byte max-reduce of "
                                    "the overflow flag (the decision to reject a step is the same on every rank)"
                                    if P > 1 else "none (P = 1: the exchange is the identity)",
        },
        "table_bytes_per_rank": 2 * (nU + nE) * Dm * 4 / P, "scaling": "weak", "parallelism": f"row-sharded x{P}",
        "backend": ("rccl" if D.backend == "nccl" else f"{D.backend} (rehearsal: collectives staged through host memory)")
        if P > 1 else None,
    }
    if P > 1:  # the fabric alone: the forward row all-to-all of one entity table, back to back
        buf = torch.empty((P * 2 * cap_e, Dm), dtype=torch.float32, device=dev)  # the entity lookup's row message (re + im)
        back = torch.empty_like(buf)
        for _ in range(3):
            parallel.all_to_all(back, buf)
        D.barrier()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            parallel.all_to_all(back, buf)
        D.barrier()
        adt = D.reduce(time.perf_counter() - t0, "max") / reps
        res["exchange"]["all_to_all_alone"] = {"message_bytes": buf.numel() * 4, "ms": adt * 1e3,
                                              "egress_GBps_per_gpu": remote * buf.numel() * 4 / adt / 1e9,
                                              "egress_GBps_per_link": remote * buf.numel() * 4 / adt / 1e9 / (P - 1),
                                              "peak_GBps_per_link": 153.0, "peak_GBps_per_gpu": 7 * 153.0,
                                              "links_used": P - 1}
    # sharded evaluation: 256 users x all items, local top-10 per rank + merge
    users = torch.from_numpy(np.random.default_rng(5).integers(1, nU, 256)).to(dev)
    hr = torch.arange(256, device=dev).repeat_interleave(50)
    hc = torch.from_numpy(np.random.default_rng(6).integers(1, nI, 256 * 50)).to(dev)
    inter = {"user_id": users}
    edt = timed(D, lambda i: sharded.full_sort_topk(inter, 10, (hr, hc)), 10, 2)
    res["full_sort_topk"] = {"users": 256, "items": nI, "k": 10, "ms": edt / 10 * 1e3,
                             "scores_per_s": 256 * nI * 10 / edt}
    del sharded, opt, model
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------------------
# configs[4]: TuckER d=200, Alibaba-iFashion shape (fp32 MFMA path)
# ---------------------------------------------------------------------------------------------------------------------
def leg_tucker(D, args):
    import numpy as np

    torch, dev = D.torch, D.dev
    from hopwise_amd.model import TuckER
    from tests.helpers import FakeDataset, make_config

    nU, nI, nE, nR, d, B = 114738, 30041, 59157, 53, 200, 2048
    cfg = make_config(dev, d, 1.0, label_smoothing=0.1, input_dropout=0.3, input_dropout1=0.4, input_dropout2=0.5)
    torch.manual_seed(0)
    m = TuckER(cfg, FakeDataset(nU, nI, nE, nR)).to(dev)
    rng = np.random.default_rng(0)
    it = {k: torch.from_numpy(v).to(dev) for k, v in dict(
        user_id=rng.integers(1, nU, B), item_id=rng.integers(1, nI, B), head_id=rng.integers(1, nE, B),
        relation_id=rng.integers(1, nR - 1, B), tail_id=rng.integers(1, nE, B)).items()}
    flops_fwd = sum(2 * B * d**3 + 2 * B * d * d + 2 * B * d * N for N in (nU + nI, nE))
    m.train()
    res = {"metric": "TuckER d=200 calculate_loss on the Alibaba-iFashion shape (rec + kg side, B=2048 each), fp32 MFMA",
           "flops_forward": flops_fwd, "unit": "TFLOP/s",
           "peak": (BF16_MFMA_PEAK_TFLOPS / 6.0 if os.environ.get("HW_GEMM_SPLIT", "1") != "0" else FP32_MFMA_PEAK_TFLOPS),
           "bound": ("bf16-mfma (3-way split)" if os.environ.get("HW_GEMM_SPLIT", "1") != "0" else "mfma"),
           "gemm": ("fp32 products as three-way bf16 splits on the bf16 matrix pipes (6 v_mfma_f32_32x32x16_bf16 per 16 k, fp32 "
                    "accumulate; error vs float64 <= the fp32 FMA chain's, tests/test_gpu_gemm_split.py)"
                    if os.environ.get("HW_GEMM_SPLIT", "1") != "0" else "v_mfma_f32_32x32x2_f32 FMA chains (HW_GEMM_SPLIT=0)"),
           "peak_note": "achieved counts the reference's fp32 FLOPs (SURVEY 8(d) config 5: 2 d^3 + 2 d^2 + 2 d N per row, x3 with "
                        "the backward); the products run as three-way bf16 splits = 6 bf16 MFMAs per fp32 MFMA-equivalent, so "
                        "the roof of the unit that executes is the dense bf16 rate / 6 = 416.7 TFLOP/s of fp32 work (the fp32-"
                        "input MFMA pipe's 157.3 TFLOP/s applies with HW_GEMM_SPLIT=0).  The kernels execute fewer FLOPs than "
                        "that count (one matrix per distinct relation instead of per row), which is why mfma_util, the cycle "
                        "fraction the matrix pipes are busy, is reported next to it"}
    peak = res["peak"]
    for mode, mult in (("forward", 1), ("forward_backward", 3)):
        def step(i):
            m.zero_grad(set_to_none=True)
            loss = m.calculate_loss(it)
            if mult == 3:
                loss.backward()
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        reps = 10
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for i in range(reps):
            step(i)
        eb.record()
        torch.cuda.synchronize()
        ms = ea.elapsed_time(eb) / reps
        res[mode] = {"ms": ms, "achieved": flops_fwd * mult / (ms * 1e-3) / 1e12,
                     "frac": flops_fwd * mult / (ms * 1e-3) / 1e12 / peak,
                     "frac_of_fp32_input_mfma_peak": flops_fwd * mult / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS}
    # full-size parity, outside the timed region: one matrix per distinct relation (the engine's default) against the
    # reference's per-row mm(r, W) on the same batch and dropout stream -- the same loss, bit for bit
    def loss_with(shared):
        m.share_relation_matrices, m._drop_calls = shared, 0
        m.input_dropout.calls = m.hidden_dropout2.calls = 0  # same Philox offsets for the [B, d] dropouts too
        with torch.no_grad():
            return float(m.calculate_loss(it))

    l_set, l_row = loss_with(True), loss_with(False)
    m.share_relation_matrices = True
    if l_set != l_row:
        raise AssertionError(f"TuckER: relation-set loss {l_set} != per-row loss {l_row}")
    res["parity"] = {"loss_relation_sets": l_set, "loss_per_row_product": l_row, "rule": "bit-identical training-mode loss"}
    mpath = os.path.join(ROOT, "profiles", "tucker_mfma_latest.json")
    if os.path.exists(mpath):
        try:
            res["mfma_util"] = json.load(open(mpath))
        except Exception:
            pass
    del m
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------------------
# configs[1], shape (i) of SURVEY §8(d): the reference's own batch layout through the model class
# ---------------------------------------------------------------------------------------------------------------------
def leg_reference_layout(D, args):
    """RotatE d=200, sample_num=256 in the layout hopwise's KnowledgeBasedDataLoader yields: rec side 8 positives x 256
    negatives tiled (2048 rows) + kg side 2048 x (1 pos + 1 neg) = 8 192 scored triples per calculate_loss, through
    hopwise_amd.model.RotatE (the drop-in class): forward, and forward + backward + dense Adam (the trainer's step)."""
    import numpy as np

    torch, dev = D.torch, D.dev
    import hopwise_amd.model as M
    from hopwise_amd.optim import FusedAdam
    from tests.helpers import FakeDataset, make_config

    nU, nI, nE, nR, K, B = 6041, 2985, 13805, 13, 256, 2048
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
    res = {"metric": "RotatE d=200 calculate_loss in the reference's batch layout (8 192 triples per step), model-class path",
           "triples_per_step": 4 * B}

    def fwd():
        with torch.no_grad():
            return m.calculate_loss(it)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.calculate_loss(it)
        loss.backward()
        opt.step()
        return loss

    for name, fn in (("forward", fwd), ("forward_backward_adam", step)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        res[name] = {"ms": dt * 1e3, "triples_per_s": 4 * B / dt}
    # the same step captured once and replayed from a HIP graph (the eager step is host-bound: ~0.1 ms of GPU work behind
    # the Python / autograd dispatch)
    from hopwise_amd.driver import GraphedTrainStep

    gstep = GraphedTrainStep(m, opt, it)
    for _ in range(10):
        gstep(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        gstep(it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    res["forward_backward_adam_graph_replay"] = {"ms": dt * 1e3, "triples_per_s": 4 * B / dt}
    del gstep
    # what a drop-in user gets under hopwise's own trainer: driver.graphed_trainer(<Trainer class>)._train_epoch over an
    # epoch of such batches, against the reference's loop (zero_grad / calculate_loss / .item() / backward / step per batch,
    # trainer.py:208-268) on the same model -- a minimal base class stands in for hopwise's Trainer on this box
    from hopwise_amd.driver import graphed_trainer

    class _Base:  # the attributes and the loop of Trainer._train_epoch that matter here
        def __init__(self, config, model):
            self.config, self.model, self.device = config, model, dev
            self.optimizer = FusedAdam(model.parameters(), lr=config["learning_rate"])
            self.clip_grad_norm = None

        def _check_nan(self, loss):
            if torch.isnan(loss):
                raise ValueError("Training loss is nan")

        def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
            self.model.train()
            total = None
            for inter in train_data:
                self.optimizer.zero_grad()
                loss = self.model.calculate_loss(inter)
                total = loss.item() if total is None else total + loss.item()
                self._check_nan(loss)
                loss.backward()
                self.optimizer.step()
            return total

    class _Batch(dict):
        def to(self, device):
            return self

    epoch = [_Batch(it) for _ in range(200)]
    cfg = {"learning_rate": 1e-3, "learner": "adam", "single_spec": True}
    for name, cls in (("reference_loop", _Base), ("graphed_trainer", graphed_trainer(_Base))):
        tr = cls(cfg, m)
        tr._train_epoch(epoch[:20], 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr._train_epoch(epoch, 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(epoch)
        res["trainer_epoch_" + name] = {"ms_per_step": dt * 1e3, "triples_per_s": 4 * B / dt}
        del tr
    res["note"] = "eager steps are host-bound; BASELINE.md's CPU probe of the reference: 0.52 M triples/s forward, 0.24 M forward+backward"
    del m, opt
    torch.cuda.empty_cache()
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, argv)
        return
    D = Dist(args)
    torch = D.torch
    out = {}
    ctx = leg_kge(D, args, out)
    if D.rank == 0:
        leg_roofline_variants(D, args, out, ctx)
        if D.world == 1 and not args.no_cpu_baseline:
            leg_cpu_baseline(D, args, out, ctx)
    traffic_tab = ctx["traffic"]
    del ctx
    torch.cuda.empty_cache()
    D.barrier()
    def guarded(name, fn):
        """A failing side leg must not take the headline line with it; the error is reported in its place (and on stderr)."""
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            import traceback

            traceback.print_exc()
            out[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            torch.cuda.empty_cache()

    if D.world == 1 and not args.no_hbm_variant:
        # right behind the RotatE variants and ahead of every leg that launches the same kernels through the model classes:
        # the profile summary splits the kernel trace by the launch log in launch order
        guarded("scorer_rooflines", lambda: leg_scorer_rooflines(D, args, traffic_tab))
        if isinstance(out.get("scorer_rooflines"), dict) and isinstance(out.get("roofline"), dict):
            out["roofline"]["other_scorers"] = out.pop("scorer_rooflines")
    if D.world == 1 and not args.no_tucker:
        guarded("configs0", lambda: leg_configs0(D, args))
        guarded("evaluation", lambda: leg_evaluation(D, args))
    if not args.no_sharded and D.world == 1:
        guarded("sharded", lambda: leg_sharded(D, args))
    if D.world == 1 and not args.no_tucker:
        guarded("tucker", lambda: leg_tucker(D, args))
        guarded("reference_layout", lambda: leg_reference_layout(D, args))
    if not args.no_pgpr:
        # second half of BASELINE.json's metric: PGPR paths expanded / s (configs[2]); every rank searches its own users
        import bench_pgpr

        torch.cuda.empty_cache()
        pg = bench_pgpr.run(D.dev, n_users=args.pgpr_users, steps=args.pgpr_steps, warmup=1,
                            cpu_users=(1 if D.rank == 0 and D.world == 1 and not args.no_cpu_baseline else 0), rank=D.rank,
                            streams=args.pgpr_streams,
                            cpu_workers=(8 if D.rank == 0 and D.world == 1 and not args.no_cpu_baseline else 0))
        if D.world > 1:  # whole-job rate on the slowest rank's clock
            tm, tp = D.reduce(pg["ms_per_step"], "max"), D.reduce(pg["paths_per_step"], "sum")
            pg.update(value=tp / (tm * 1e-3), ms_per_step=tm, paths_per_step=tp, n_gpus=D.world)
        out["pgpr"] = pg
    if not args.no_cafe and D.world == 1:
        # "PGPR / CAFE beam-search traversal": the CAFE program executor on the same graph (bench_cafe.py)
        import bench_cafe

        torch.cuda.empty_cache()
        # 1 024 users per call: one workgroup per user, four per CU (256 per call leave three quarters of the chip idle)
        guarded("cafe", lambda: bench_cafe.run(D.dev, n_users=args.cafe_users,
                                               cpu_users=(2 if not args.no_cpu_baseline else 0), streams=args.cafe_streams))
    if not args.no_sharded and D.world > 1:
        # The one leg whose ranks talk to each other (all-to-alls, all-reduces) runs LAST at N > 1, under a deadline: every
        # number above is a replica measurement that needs no peer, and a collective that never completes (a rank that died,
        # a fabric problem) must not take them along.  Each rank arms the same timer; when it fires rank 0 prints the line
        # with the leg marked as timed out and every rank leaves at once.
        import threading

        finished = threading.Event()

        def on_deadline():
            if finished.wait(args.sharded_deadline):
                return
            if D.rank == 0:
                out["sharded"] = {"error": f"timed out after {args.sharded_deadline:.0f} s inside the row-sharded leg (a "
                                           "collective did not complete); the replica legs above are unaffected"}
                print(json.dumps(out), flush=True)
            sys.stdout.flush()
            # a collective that never completed is a FAILED run: every rank leaves with a non-zero code (the line, with the leg
            # marked as timed out, is already on rank 0's stdout; launch_ranks relays it whatever the exit code)
            os._exit(3)

        threading.Thread(target=on_deadline, daemon=True).start()
        torch.cuda.empty_cache()
        guarded("sharded", lambda: leg_sharded(D, args))
        finished.set()
    if args.workload == "sharded" and "value" in out.get("sharded", {}):
        sh = out["sharded"]
        out["replica_configs1"] = {k: out[k] for k in ("metric", "value", "unit", "ms_per_step", "config")}
        out.update(metric=sh["metric"], value=sh["value"], unit=sh["unit"], steps=sh["steps"], warmup=sh["warmup"],
                   ms_per_step=sh["ms_per_step"],
                   config={"workload": "configs[3]: ComplEx dim=256, Yelp2018-shaped KG, user/entity tables row-sharded",
                           "parallelism": sh["parallelism"], "global_batch_triples": sh["batch_per_rank"]["triples"] * D.world})
    if D.rank == 0:
        if os.environ.get("HW_BENCH_LAUNCH_LOG"):
            with open(os.environ["HW_BENCH_LAUNCH_LOG"], "w") as f:
                json.dump(LAUNCH_LOG, f, indent=1)
        print(json.dumps(out))
    D.close()


if __name__ == "__main__":
    main()
