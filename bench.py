#!/usr/bin/env python
"""Headline benchmark: KGE triples scored / s on BASELINE.json configs[1]
(RotatE dim=200, 256 negatives per positive, ml-1m-small-shaped synthetic KG), per GPU and whole job.

One *step* = one pass of the scoring hot path over one batch, inputs resident in HBM:
    1. device negative sampling: 256 negatives for each of 2048 (user, positive item) pairs, rejecting
       the user's interacted items (hw_neg_sample, writes straight into the grouped index buffer);
    2. grouped RotatE scorer over 2048 x (1 + 256) = 526 336 (h, r, t) triples (hw_kge_score_grouped);
    3. self-adversarial negative-weighting loss over the [2048, 257] scores (hw_loss_self_adversarial).
value = triples scored per second over the whole job (all ranks).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the 22 MB of tables are replicated
and every rank scores its own batches -- the path shards by independent (h, r) groups, there is no
data-path collective, so scaling is weak.  The row-sharded all-to-all lookup for tables
that are worth sharding is exercised by tests/test_dist_gloo.py, not by this default run.

The JSON line also carries `roofline` for the dominant kernel (the grouped scorer, HBM-bound) and
`cpu_baseline` (oracle/kge_ref.c, the C restatement, on the host cores).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DIM, K_NEG, B_POS = 200, 256, 2048
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_triple(dim=DIM, k=K_NEG):
    """tail rows 2*d*4 + tail id 8 + score 4 + shared (h_re, h_im, theta rows + 2 ids)/(K+1)."""
    return 2 * dim * 4 + 8 + 4 + (3 * dim * 4 + 16) / (k + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-variant", action="store_true")
    ap.add_argument("--no-pgpr", action="store_true")
    ap.add_argument("--unfused-sampler", action="store_true", help="run hw_neg_sample as its own launch before the scorer")
    ap.add_argument("--pgpr-steps", type=int, default=5)
    ap.add_argument("--pgpr-users", type=int, default=256,
                    help="users per beam search: 64 -> 5.8 G paths/s, 256 -> 6.7 G (the first hops' fixed cost amortised), ~8 GB of work buffers")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--entities", type=int, default=0, help="override n_entities (HBM-resident variant, e.g. 2600000)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # backend "nccl" is RCCL on ROCm; HW_BENCH_BACKEND=gloo exists only to rehearse the multi-process control flow
        # with several ranks sharing one GPU (RCCL refuses two ranks on one device)
        dist.init_process_group(backend=os.environ.get("HW_BENCH_BACKEND", "nccl"))
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)

    from hopwise_amd import ops
    from hopwise_amd.data import synthetic

    kg = synthetic.make("ml-1m-small", seed=2024)
    nE = args.entities if args.entities > 0 else kg.n_entities
    g = torch.Generator(device="cpu").manual_seed(2024)

    def xavier(rows):
        return (torch.randn(rows, DIM, generator=g) * (2.0 / (rows + DIM)) ** 0.5).to(dev)

    user_re, user_im = xavier(kg.n_users), xavier(kg.n_users)
    ent_re, ent_im = xavier(nE), xavier(nE)
    theta = (torch.rand(kg.n_relations, DIM, generator=g) * (2 * np.pi)).to(dev)
    head, rel, tail = (user_re, user_im), (theta,), (ent_re, ent_im)
    # rec side: tails are items, i.e. entity rows [0, n_items) (what the model class passes for `side == "rec"`)
    desc = ops.make_desc("RotatE", head, rel, tail, margin=1.0, tail_rows=(kg.n_items if args.entities == 0 else None))

    # batches differ per rank; sampler "used" sets = the user's interacted items
    rng = np.random.default_rng(2024 + rank)
    pick = rng.integers(0, len(kg.inter_users), B_POS)
    users = torch.from_numpy(kg.inter_users[pick]).to(dev)
    pos = torch.from_numpy(kg.inter_items[pick]).to(dev)
    uptr, uidx = synthetic.used_csr(kg.inter_users, kg.inter_items, kg.n_users)
    uptr, uidx = torch.from_numpy(uptr).to(dev), torch.from_numpy(uidx).to(dev)
    n_items = kg.n_items if args.entities == 0 else nE  # HBM variant draws negatives over the whole big table
    t_idx = torch.empty((B_POS, 1 + K_NEG), dtype=torch.int64, device=dev)  # one row of tail ids per (h, r) group
    t_idx[:, 0] = pos
    neg_view = t_idx[:, 1:]  # the sampler writes its [keys, draws] output straight into the grouped buffer
    ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    # The step is four short launches (~75 us of GPU work), so the host must not be the bottleneck: outputs are
    # allocated once and the C-ABI entry points are called directly with pre-built argument tuples (the Philox offset
    # is the only argument that changes per step).
    import ctypes as C

    from hopwise_amd import _lib

    lib = _lib.load()
    scores = torch.empty((B_POS, 1 + K_NEG), dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.hw_loss_workspace_floats()) + B_POS, dtype=torch.float32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    samp_head = (P(uptr), P(uidx), uptr.numel() - 1, P(users), B_POS, K_NEG, n_items, None, None, None, 0,
                 C.c_uint64(2024 + rank))
    samp_tail = (P(neg_view), neg_view.stride(0), neg_view.stride(1), P(oob), stream)
    score_args = (C.byref(desc), P(users), None, kg.ui_relation, P(t_idx), t_idx.stride(0), t_idx.stride(1), B_POS,
                  1 + K_NEG, P(scores), scores.stride(0), scores.stride(1), P(oob), stream)
    loss_args = (P(scores), B_POS, 1 + K_NEG, C.c_float(1.0), P(loss), None, P(ws), stream)

    # Negatives are drawn inside the scorer when the item space fits its per-wave LDS bitmap (hw_kge_score_grouped_sampled:
    # the same ids hw_neg_sample would write, one launch less, the id matrix written but never re-read); otherwise the
    # sampler runs as its own launch first.
    fused = n_items <= 65536 and not args.unfused_sampler
    fused_args = (C.byref(desc), P(users), None, kg.ui_relation, P(t_idx), t_idx.stride(0), t_idx.stride(1), B_POS,
                  1 + K_NEG, P(uptr), P(uidx), uptr.numel() - 1, P(users), n_items, C.c_uint64(2024 + rank))
    fused_tail = (P(scores), scores.stride(0), scores.stride(1), P(oob), stream)

    def step(i, timed=False):
        rc = 0
        if not fused:
            rc = lib.hw_neg_sample(*samp_head, C.c_uint64(i), *samp_tail)
        if timed:
            ev_a[i].record()
        if fused:
            rc |= lib.hw_kge_score_grouped_sampled(*fused_args, C.c_uint64(i), *fused_tail)
        else:
            rc |= lib.hw_kge_score_grouped(*score_args)
        if timed:
            ev_b[i].record()
        rc |= lib.hw_loss_self_adversarial(*loss_args)
        if rc:
            raise RuntimeError(f"launch failed: {rc}")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, timed=True)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=(dev if dist.get_backend() == "nccl" else "cpu"))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(loss).all().item()
    assert int(oob.item()) == 0 and ops.oob_count(dev) == 0

    triples_per_step = B_POS * (1 + K_NEG)
    value = world * triples_per_step * args.steps / dt
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_a, ev_b)]))
    abytes = algorithmic_bytes_per_triple() * triples_per_step
    achieved = abytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("kge_score_grouped_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "KGE triples scored per second (RotatE d=200, 256 negatives; sampler + scorer + self-adversarial loss)",
        "value": value,
        "unit": "triples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: RotatE dim=200, 256 negatives, ml-1m-small-shaped KG, grouped scorer B=2048 x 257",
            "n_users": kg.n_users, "n_items": kg.n_items, "n_entities": nE, "n_relations": kg.n_relations,
            "triples_per_step": triples_per_step, "parallelism": f"replicas x{world}",
        },
        "roofline": {
            "bound": "hbm", "kernel": "kge_score_grouped_kernel<RotatE" + (", SAMPLE>" if fused else ">"), "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": abytes,
            "scorer_only_triples_per_s": triples_per_step / (kern_ms * 1e-3),
            "note": "configs[1]'s tables (22 MB) stay resident in L2 / Infinity Cache, so algorithmic GB/s can exceed the "
                    "HBM peak and `traffic` (PMC) is far below the algorithmic bytes; hbm_resident_variant is the same "
                    "kernel on a 4.2 GB table, where every tail row comes from HBM",
        },
    }

    if rank == 0 and world == 1 and not args.no_hbm_variant:
        # Same kernel, same batch shape, but an entity table far beyond L2 + Infinity Cache (2.6 M rows x 2 tables
        # x 800 B = 4.2 GB) with negatives drawn over all of it: every tail row is an HBM miss.
        nbig = 2_621_440
        big_re = torch.randn(nbig, DIM, device=dev) * 0.01
        big_im = torch.randn(nbig, DIM, device=dev) * 0.01
        bdesc = ops.make_desc("RotatE", head, rel, (big_re, big_im), margin=1.0)
        bidx = torch.randint(1, nbig, (B_POS, 1 + K_NEG), device=dev, dtype=torch.int64)
        for _ in range(3):
            ops.kge_score_grouped_raw(bdesc, users, None, kg.ui_relation, bidx)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        ea.record()
        for _ in range(reps):
            ops.kge_score_grouped_raw(bdesc, users, None, kg.ui_relation, bidx)
        eb.record()
        torch.cuda.synchronize()
        bms = ea.elapsed_time(eb) / reps
        out["roofline"]["hbm_resident_variant"] = {
            "n_entities": nbig, "table_bytes": 2 * nbig * DIM * 4, "kernel_ms": bms,
            "achieved": abytes / (bms * 1e-3) / 1e9, "frac": abytes / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "triples_per_s": triples_per_step / (bms * 1e-3),
        }
        del big_re, big_im, bidx

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cref

        tabs = [t.cpu().numpy() for t in (user_re, user_im, theta, ent_re, ent_im)]
        h_np, t_np = users.cpu().numpy(), np.ascontiguousarray(t_idx.cpu().numpy())
        outb = np.empty(t_np.shape, np.float32)
        cref.score_grouped("RotatE", tabs[0:2], tabs[2:3], tabs[3:5], h_np, None, kg.ui_relation, t_np, 1.0, 0.0, outb)
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds:
            cref.score_grouped("RotatE", tabs[0:2], tabs[2:3], tabs[3:5], h_np, None, kg.ui_relation, t_np, 1.0, 0.0, outb)
            reps += 1
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": reps * triples_per_step / cdt, "unit": "triples/s", "cores": cref.threads(), "kind": "port",
            "sample": f"{reps} passes of the same 2048 x 257 grouped RotatE scoring batch (scorer only) through "
                      f"oracle/kge_ref.c (OpenMP) in {cdt:.1f} s; host has {os.cpu_count()} logical CPUs",
        }
        gpu_scores = ops.kge_score_grouped_raw(desc, users, None, kg.ui_relation, t_idx).cpu().numpy()
        np.testing.assert_allclose(gpu_scores, outb, rtol=1e-4, atol=1e-5)  # same inputs, same answers
    if not args.no_pgpr:
        # second half of BASELINE.json's metric: PGPR paths expanded / s (configs[2]); every rank searches its own users
        import bench_pgpr

        torch.cuda.empty_cache()
        pg = bench_pgpr.run(dev, n_users=args.pgpr_users, steps=args.pgpr_steps, warmup=1,
                            cpu_users=(1 if rank == 0 and world == 1 and not args.no_cpu_baseline else 0), rank=rank)
        if world > 1:
            rdev = dev if dist.get_backend() == "nccl" else "cpu"
            tm = torch.tensor([pg["ms_per_step"]], device=rdev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            # whole-job rate on the slowest rank's clock
            tp = torch.tensor([pg["paths_per_step"]], device=rdev, dtype=torch.float64)
            dist.all_reduce(tp, op=dist.ReduceOp.SUM)
            pg["value"] = float(tp.item() / (tm.item() * 1e-3))
            pg["ms_per_step"] = float(tm.item())
            pg["paths_per_step"] = float(tp.item())
        out["pgpr"] = pg
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
