#!/usr/bin/env python
"""Full sort and evaluation step (scores -> history mask -> top-10) of the distance scorers (RotatE d = 200, TransE d = 64 / 200)
as one product on the MFMA GEMM + finish pass + exact re-rank (model.gemm_full_sort, the default) against the gather kernels,
at ml-1m-small's 2 985 items and Yelp2018's 45 539."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hopwise_amd.model as M
from hopwise_amd import ops
from tests.helpers import FakeDataset, make_config

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for nU, nI, nE, nR in ((6041, 2985, 13805, 13), (45920, 45539, 90962, 44)):
    for name, d in (("RotatE", 200), ("TransE", 200), ("TransE", 64)):
        torch.manual_seed(0)
        m = getattr(M, name)(make_config(dev, d, 1.0), FakeDataset(nU, nI, nE, nR)).to(dev)
        m.eval()
        rng = np.random.default_rng(0)
        for B in (256, 2048):
            users = torch.from_numpy(rng.integers(1, nU, B)).to(dev)
            hist = (torch.from_numpy(rng.integers(0, B, 50 * B)).to(dev), torch.from_numpy(rng.integers(1, nI, 50 * B)).to(dev))
            inter = {"user_id": users}
            out = {"model": name, "d": d, "B": B, "N": nI}
            with torch.no_grad():
                for tag, flag in (("gemm", True), ("kernel", False)):
                    m.gemm_full_sort = flag
                    out[f"full_sort_{tag}_ms"] = round(timed(lambda: m.full_sort_predict(inter)), 4)
                    out[f"eval_top10_{tag}_ms"] = round(timed(lambda: m.full_sort_topk(inter, 10, hist)), 4)
                m.gemm_full_sort = True
                m.gemm_rerank = False
                out["eval_top10_gemm_norerank_ms"] = round(timed(lambda: m.full_sort_topk(inter, 10, hist)), 4)
                m.gemm_rerank = True
                gv, gi = m.full_sort_topk(inter, 10, hist)
                a = m.full_sort_predict(inter)
                # stages of the product form
                head, rel, tail = m._tables("rec")
                q, qn = ops.kge_query_rows(name, head, rel, tail, users, None, m._rec_rel_full_sort(), sqnorm=True, margin=m.margin)
                table, tn = m._cat_tables(tail)[:nI], m._tail_sqnorm(tail)[:nI]
                buf = torch.empty((B, (nI + 63) // 64 * 64), dtype=torch.float32, device=dev)[:, :nI]
                out["query_rows_ms"] = round(timed(lambda: ops.kge_query_rows(name, head, rel, tail, users, None, m._rec_rel_full_sort(), sqnorm=True)), 4)
                out["gemm_ms"] = round(timed(lambda: ops.gemm(q, table, out=buf)), 4)
                out["finish_ms"] = round(timed(lambda: ops.kge_dist_finish_(buf, q, table, qn, tn, 0.0)), 4)
                out["gemm_dist_ms"] = round(timed(lambda: ops.gemm_dist(q, table, qn, tn, 0.0, out=buf)), 4)
                out["topk32_ms"] = round(timed(lambda: ops.topk_rows(buf, 32)), 4)
                out["topk10_ms"] = round(timed(lambda: ops.topk_rows(buf, 10)), 4)
                m.gemm_full_sort = False
                b = m.full_sort_predict(inter).view_as(a)
                kv, ki = m.full_sort_topk(inter, 10, hist)
                base = m.margin if name == "RotatE" else 0.0
                out["max_rel_err_of_distance"] = float(((a - b).abs() / (base - b).abs().clamp_min(1e-6)).max())
                out["topk_indices_equal"] = float((gi == ki).float().mean())
            print(json.dumps(out), flush=True)
