#!/usr/bin/env python
"""The evaluation's ranking product three ways: six split terms (hw_gemm_f32), three terms on the 128 x 128 tiles (hw_gemm_rank),
three terms from plane images on 256 x 256 tiles (hw_gemm_rank_planes); with the mask / stretch-maximum by-products."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hopwise_amd import ops
dev = torch.device("cuda", 0)
def t_us(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K in ((2048, 45539, 400), (2048, 45539, 200), (2048, 45539, 64), (256, 45539, 400), (2048, 2985, 400)):
    q, t, b = torch.randn((M, K), device=dev), torch.randn((N, K), device=dev), torch.randn(N, device=dev)
    out = ops.padded_rows(M, N, dev)
    bits = ops.mask_bits(None, M, N, device=dev)
    tm = torch.empty((M, (N + 63) // 64), device=dev)
    qp, tp = ops.rank_planes(q), ops.rank_planes(t)
    r = {"M": M, "N": N, "K": K,
         "six_terms_us": round(t_us(lambda: ops.gemm(q, t, bias=b, out=out)), 1),
         "three_terms_us": round(t_us(lambda: ops.gemm_rank(q, t, b, out=out, mask=bits, tile_max=tm)), 1),
         "planes_us": round(t_us(lambda: ops.gemm_rank_planes(qp, tp, M, N, K, b, out=out, mask=bits, tile_max=tm)), 1),
         "planes_plain_us": round(t_us(lambda: ops.gemm_rank_planes(qp, tp, M, N, K, None, out=out)), 1),
         "planes_max_only_us": round(t_us(lambda: ops.gemm_rank_planes(qp, tp, M, N, K, b, out=out, tile_max=tm)), 1),
         "three_terms_plain_us": round(t_us(lambda: ops.gemm_rank(q, t, None, out=out)), 1),
         "query_planes_us": round(t_us(lambda: ops.rank_planes(q)), 1), "table_planes_us": round(t_us(lambda: ops.rank_planes(t)), 1)}
    print(json.dumps(r), flush=True)
