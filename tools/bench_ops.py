#!/usr/bin/env python
"""Per-kernel micro-benchmark at the BASELINE config-2 shapes (T5-base, B=16, N=20, L=200, T=8).
Random (not zero) operands; every op timed with HIP events over `--iters` back-to-back launches after a warm-up.
    python tools/bench_ops.py [--only gemm,attn,norm] [--iters 20]
Prints one line per op: avg µs, TFLOP/s (MFMA ops) or GB/s of algorithmic bytes (HBM ops)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lako_amd.ops import HipOps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="gemm,tn,attn,norm,misc,index,retriever")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--dephase", default="0", help="gemm_nt_dephase values (10-ns ticks) to time, e.g. 0,500,1000")
ap.add_argument("--variants", default="-1", help="gemm_nt tile variants to time, e.g. 0,1,2")
ap.add_argument("--rows", type=int, default=0, help="encoder token rows of the GEMM shapes (default B*N*L = 64000; 47757 = the bench's valid tokens)")
ap.add_argument("--vendor", action="store_true", help="also time torch.matmul (hipBLASLt / rocBLAS) on the GEMM shapes: a yardstick, "
                "never part of the product path")
args = ap.parse_args()
only = set(args.only.split(","))
T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
ES = 2 if T == torch.bfloat16 else 4
dev = torch.device("cuda:0")
ops = HipOps()
B, N, L, Tt, d, f, H, dk, V, Ld = 16, 20, 200, 8, 768, 3072, 12, 64, 32128, 12
inner, Me, Md, S = H * dk, B * N * L, B * Tt, N * L
if args.rows:
    Me = args.rows


def rnd(*shape, dtype=T, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def timeit(name, fn, flops=0.0, bytes_=0.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.iters
    extra = f"{flops / us / 1e6:9.1f} TFLOP/s" if flops else (f"{bytes_ / us / 1e3:9.1f} GB/s" if bytes_ else "")
    print(f"{name:44s} {us:10.1f} us  {extra}", flush=True)


drop = (0.1, 1, 2)
for variant, dephase in ([(int(v), int(dp)) for v in args.variants.split(",") for dp in args.dephase.split(",")] if "gemm" in only else []):
    ops.set_tuning("gemm_nt_dephase", dephase % 100000)
    ops.set_tuning("gemm_nt_dephase_n", max(2, dephase // 100000))
    print(f"--- dephase {dephase}")
    ops.set_tuning("gemm_nt_variant", variant if variant < 0 else variant % 10)
    ops.set_tuning("gemm_nt_persistent", 0 if 10 <= variant < 20 else 1)
    ops.set_tuning("gemm_nt_stagger", 0 if 20 <= variant < 30 else 1)
    ops.set_tuning("gemm_nt_wide_epi", 0 if 30 <= variant < 40 else 1)
    dbg = {7: 1, 8: 2, 9: 4, 14: 8, 15: 16, 11: 128 << 8, 12: 64 << 8, 13: (128 << 8) | 1}.get(variant // 10, 0)   # 7x: no K-loop DMA; 8x: every DMA hits L2 (timing experiments, wrong results)
    if dbg or os.environ.get("LAKO_LIB"):      # (the release library refuses the key: experiments need LAKO_LIB=…/liblako_hip_exp.so)
        ops.set_tuning("gemm_nt_debug", dbg)
    ops.set_tuning("gemm_nt_group_m", {4: 0, 5: 4, 6: 16}.get(variant // 10, 8))
    print(f"--- gemm_nt variant {variant if variant < 0 else variant % 10} (-1 auto, 0 128x128, 1 256x128, 2 256x256 / 8 waves, 4 ring128, 6 256x256 / 4 waves) "
          f"persistent={not 10 <= variant < 20} stagger={not 20 <= variant < 30} wide_epi={not 30 <= variant < 40} "
          f"group_m={ {4: 0, 5: 4, 6: 16}.get(variant // 10, 8)}", flush=True)
    for nm, (M, Nn, K), kw in [
        ("nt qkv   [Me,768]x[2304,768]", (Me, 3 * inner, d), {}),
        ("nt o+res [Me,768]x[768,768]", (Me, d, inner), dict(resid=True, drop=drop)),
        ("nt wi    [Me,768]x[3072,768] relu+drop", (Me, f, d), dict(relu=True, drop=drop)),
        ("nt wo    [Me,3072]x[768,3072] res+drop", (Me, d, f), dict(resid=True, drop=drop)),
        ("nt plain [Me,768]x[768,768]", (Me, d, inner), {}),
        ("nt plain [Me,768]x[3072,768]", (Me, f, d), {}),
        ("nt plain [Me,3072]x[768,3072]", (Me, d, f), {}),
        ("nt plain [Me,2304]x[768,2304] (dX of QKV)", (Me, d, 3 * inner), {}),
        ("nt kvall [Me,768]x[18432,768]", (Me, Ld * 2 * inner, d), {}),
        ("nt dxkv  [Me,18432]x[768,18432]", (Me, d, Ld * 2 * inner), {}),
        ("nt dpre  [Me,768]x[3072,768] auxmask", (Me, f, d), dict(aux=True)),
        ("nt lmhead[128,768]x[32128,768] f32 out", (Md, V, d), dict(f32=True)),
        ("nt dec qkv [128,768]x[2304,768]", (Md, 3 * inner, d), {}),
        ("nt dec o+res [128,768]x[768,768]", (Md, d, inner), dict(resid=True, drop=drop)),
        ("nt dec wi  [128,768]x[3072,768] relu+drop", (Md, f, d), dict(relu=True, drop=drop)),
        ("nt dec wo  [128,3072]x[768,3072] res+drop", (Md, d, f), dict(resid=True, drop=drop)),
        ("nt square 4096^3", (4096, 4096, 4096), {}),
        ("nt square 8192^3", (8192, 8192, 8192), {}),
        ("nt square 8192^3, row stride 8192+64", (8192, 8192, 8192), dict(pad=64)),
        ("nt kvall, row stride 768+64", (Me, Ld * 2 * inner, d), dict(pad=64)),
    ]:
        pad = kw.get("pad", 0)       # row stride K + pad elements: breaks power-of-two strides (L2 channel spread)
        A, Bm = rnd(M, K + pad)[:, :K], rnd(Nn, K + pad)[:, :K]
        C = torch.empty(M, Nn, dtype=torch.float32 if kw.get("f32") else T, device=dev)
        k2 = {}
        if kw.get("resid"):
            k2["resid"] = rnd(M, Nn)
        if kw.get("aux"):
            k2["aux"], k2["aux_scale"] = rnd(M, Nn), 1.1
        if kw.get("relu"):
            k2["relu"] = True
        if kw.get("drop"):
            k2["drop"] = kw["drop"]
        timeit(nm, lambda: ops.gemm_nt(A, Bm, C, **k2), flops=2.0 * M * Nn * K)
        del A, Bm, C, k2

if args.vendor:
    print("--- vendor yardstick: torch.matmul, plain bf16 GEMM without any fused epilogue", flush=True)
    for nm, (M, Nn, K) in [("nt [Me,768]x[2304,768]", (Me, 3 * inner, d)), ("nt [Me,768]x[768,768]", (Me, d, inner)),
                           ("nt [Me,768]x[3072,768]", (Me, f, d)), ("nt [Me,3072]x[768,3072]", (Me, d, f)),
                           ("nt [Me,2304]x[768,2304]", (Me, d, 3 * inner)),
                           ("nt [Me,768]x[18432,768]", (Me, Ld * 2 * inner, d)), ("nt [Me,18432]x[768,18432]", (Me, d, Ld * 2 * inner)),
                           ("nt 4096^3", (4096,) * 3), ("nt 8192^3", (8192,) * 3)]:
        A, Bm = rnd(M, K), rnd(Nn, K)
        C = torch.empty(M, Nn, dtype=T, device=dev)
        timeit("vendor " + nm, lambda: torch.matmul(A, Bm.t(), out=C), flops=2.0 * M * Nn * K)
        del A, Bm, C
    for nm, (K, M, Nn) in [("tn [Me,2304]^T x [Me,768]", (Me, 3 * inner, d)), ("tn [Me,768]^T x [Me,768]", (Me, d, inner)),
                           ("tn [Me,3072]^T x [Me,768]", (Me, f, d)), ("tn [Me,18432]^T x [Me,768]", (Me, Ld * 2 * inner, d))]:
        A, Bm = rnd(K, M), rnd(K, Nn)
        C = torch.empty(M, Nn, dtype=T, device=dev)
        timeit("vendor " + nm, lambda: torch.matmul(A.t(), Bm, out=C), flops=2.0 * M * Nn * K)
        del A, Bm, C

# (gemm_tn_big = 2, the accumulation switched off, exists in the experiments build only: LAKO_LIB=…/liblako_hip_exp.so)
for big in (([1, 2, 0] if os.environ.get("LAKO_LIB") else [1, 0]) if "tn" in only else []):
    ops.set_tuning("gemm_tn_big", big)
    print(f"--- gemm_tn 256x256 kernel {['off (128x128)', 'on', 'on, accumulation atomics skipped (timing experiment)'][big]}", flush=True)
    for nm, (K, M, Nn) in [
        ("tn dWqkv [Me,2304]^T x [Me,768]", (Me, 3 * inner, d)),
        ("tn dWo   [Me,768]^T x [Me,768]", (Me, d, inner)),
        ("tn dWi   [Me,3072]^T x [Me,768]", (Me, f, d)),
        ("tn dWo2  [Me,768]^T x [Me,3072]", (Me, d, f)),
        ("tn dWkv  [Me,18432]^T x [Me,768]", (Me, Ld * 2 * inner, d)),
        ("tn dE    [128,32128]^T x [128,768]", (Md, V, d)),
    ]:
        A, Bm = rnd(K, M), rnd(K, Nn)
        C = torch.zeros(M, Nn, device=dev)
        timeit(nm, lambda: ops.gemm_tn(A, Bm, C), flops=2.0 * M * Nn * K)
        del A, Bm, C

if "tn" in only:
    print("--- gemm_tn grouped: the four weight gradients of an encoder layer in one launch", flush=True)
    dys = [rnd(Me, 3 * inner), rnd(Me, d), rnd(Me, f), rnd(Me, d)]
    xs = [rnd(Me, d), rnd(Me, inner), rnd(Me, d), rnd(Me, f)]
    cs = [torch.zeros(a.shape[1], b.shape[1], device=dev) for a, b in zip(dys, xs)]
    fl4 = sum(2.0 * Me * a.shape[1] * b.shape[1] for a, b in zip(dys, xs))
    for big in (1, 2):
        ops.set_tuning("gemm_tn_big", big)
        timeit(f"tn layer: 4 separate launches{' (no atomics)' if big == 2 else ''}", lambda: [ops.gemm_tn(a, b, c) for a, b, c in zip(dys, xs, cs)], flops=fl4)
        timeit(f"tn layer: 1 grouped launch{' (no atomics)' if big == 2 else ''}", lambda: ops.gemm_tn_grouped([(a, b, c, 1.0) for a, b, c in zip(dys, xs, cs)]), flops=fl4)
    ops.set_tuning("gemm_tn_big", 1)
    A, Bm = rnd(Me, Ld * 2 * inner), rnd(Me, d)
    Ck = torch.zeros(Ld * 2 * inner, d, device=dev)
    for sp in (0, 2, 3, 5, 7, 9, 14):
        ops.set_tuning("gemm_tn_split", sp)
        timeit(f"tn layer grouped, K-splits {sp or 'auto'}", lambda: ops.gemm_tn_grouped([(a, b, c, 1.0) for a, b, c in zip(dys, xs, cs)]), flops=fl4)
        timeit(f"tn dWkv, K-splits {sp or 'auto'}", lambda: ops.gemm_tn(A, Bm, Ck), flops=2.0 * Me * Ld * 2 * inner * d)
    ops.set_tuning("gemm_tn_split", 0)
    del dys, xs, cs, A, Bm, Ck

if "attn" in only:
    BN = B * N
    qkv = rnd(BN * L, 3 * inner, scale=0.5)
    ctx = torch.empty(BN * L, inner, dtype=T, device=dev)

    def heads(t, rb, rt, c0):
        return t.view(rb, rt, t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))
    st = torch.empty(BN, H, L, 4, device=dev)
    rel = rnd(H, 2 * L - 1, dtype=torch.float32)
    lens = torch.randint(L // 2, L + 1, (BN,), device=dev)
    km = (torch.arange(L, device=dev)[None] < lens[:, None]).to(torch.uint8)
    fl = 4.0 * BN * H * L * L * dk
    for nm, kw in [("attn_fwd enc plain", {}), ("attn_fwd enc bias+mask", dict(rel_bias=rel, rel_off=L - 1, key_mask=km)),
                   ("attn_fwd enc bias+mask+drop", dict(rel_bias=rel, rel_off=L - 1, key_mask=km, drop=drop))]:
        timeit(nm, lambda: ops.attn_fwd(heads(qkv, BN, L, 0), heads(qkv, BN, L, inner), heads(qkv, BN, L, 2 * inner),
                                        heads(ctx, BN, L, 0), st, **kw), flops=fl)
    ops.attn_fwd(heads(qkv, BN, L, 0), heads(qkv, BN, L, inner), heads(qkv, BN, L, 2 * inner), heads(ctx, BN, L, 0), st,
                 rel_bias=rel, rel_off=L - 1, key_mask=km, drop=drop)
    dctx, dqkv = rnd(BN * L, inner), torch.empty(BN * L, 3 * inner, dtype=T, device=dev)
    drel = torch.zeros_like(rel)
    bm = dict(rel_bias=rel, rel_off=L - 1, key_mask=km)
    for nm, kw in [("attn_bwd enc plain", {}), ("attn_bwd enc bias+mask", bm), ("attn_bwd enc bias+mask+drop", dict(bm, drop=drop)),
                   ("attn_bwd enc bias+mask+drel", dict(bm, drel=drel)),
                   ("attn_bwd enc bias+mask+drop+drel", dict(bm, drop=drop, drel=drel))]:
        timeit(nm, lambda: ops.attn_bwd(heads(qkv, BN, L, 0), heads(qkv, BN, L, inner), heads(qkv, BN, L, 2 * inner),
                                        heads(ctx, BN, L, 0), heads(dctx, BN, L, 0), st, heads(dqkv, BN, L, 0),
                                        heads(dqkv, BN, L, inner), heads(dqkv, BN, L, 2 * inner), **kw), flops=2.5 * fl)
    # the same encoder attention on RAGGED rows (only the valid tokens, packed) — what the unpadded engine path launches
    offs = torch.zeros(BN + 1, dtype=torch.int32, device=dev)
    offs[1:] = torch.cumsum(lens, 0)
    Mv = int(offs[-1])
    sel = km.bool().reshape(-1)
    qkv_r = qkv[sel].contiguous()
    ctx_r, dctx_r, dqkv_r = torch.empty(Mv, inner, dtype=T, device=dev), dctx[sel].contiguous(), torch.empty(Mv, 3 * inner, dtype=T, device=dev)

    def hr(t, c0):
        return t.view(1, t.shape[0], t.shape[1])[:, :, c0:c0 + inner].unflatten(2, (H, dk))
    rg = dict(q_off=offs, k_off=offs, max_q=L, max_k=L, rel_bias=rel, rel_off=L - 1)
    flr = fl * Mv / (BN * L)
    timeit(f"attn_fwd enc RAGGED bias+drop ({Mv} of {BN * L} rows)", lambda: ops.attn_fwd(hr(qkv_r, 0), hr(qkv_r, inner), hr(qkv_r, 2 * inner), hr(ctx_r, 0), st, drop=drop, **rg), flops=flr)
    for nm, kw in [("attn_bwd enc RAGGED bias+drop", dict(drop=drop)), ("attn_bwd enc RAGGED bias+drop+drel", dict(drop=drop, drel=drel))]:
        timeit(nm, lambda: ops.attn_bwd(hr(qkv_r, 0), hr(qkv_r, inner), hr(qkv_r, 2 * inner), hr(ctx_r, 0), hr(dctx_r, 0), st, hr(dqkv_r, 0),
                                        hr(dqkv_r, inner), hr(dqkv_r, 2 * inner), **rg, **kw), flops=2.5 * flr)
    # cross attention: T=8 queries over S=4000 keys
    kv = rnd(B * S, 2 * inner, scale=0.5)
    q = rnd(Md, inner, scale=0.5)
    c2 = torch.empty(Md, inner, dtype=T, device=dev)
    st2 = torch.empty(B, H, Tt, 4, device=dev)
    em = km.view(B, S)
    flc = 4.0 * B * H * Tt * S * dk
    byc = 2.0 * B * S * inner * ES
    timeit("attn_fwd cross (T=8,S=4000)", lambda: ops.attn_fwd(heads(q, B, Tt, 0), heads(kv, B, S, 0), heads(kv, B, S, inner),
                                                                heads(c2, B, Tt, 0), st2, key_mask=em, drop=drop), bytes_=byc)
    dkv, dq, dc2 = torch.empty_like(kv), torch.empty_like(q), rnd(Md, inner)
    timeit("attn_bwd cross", lambda: ops.attn_bwd(heads(q, B, Tt, 0), heads(kv, B, S, 0), heads(kv, B, S, inner),
                                                  heads(c2, B, Tt, 0), heads(dc2, B, Tt, 0), st2, heads(dq, B, Tt, 0),
                                                  heads(dkv, B, S, 0), heads(dkv, B, S, inner), key_mask=em, drop=drop),
           bytes_=2 * byc)

if "norm" in only:
    x, w = rnd(Me, d), torch.ones(d, device=dev)
    y, rs = torch.empty_like(x), torch.empty(Me, device=dev)
    timeit("rmsnorm_fwd [Me,768]", lambda: ops.rmsnorm_fwd(x, w, y, rs, 1e-6), bytes_=2.0 * Me * d * ES)
    dy, dx, dw = rnd(Me, d), torch.empty_like(x), torch.zeros(d, device=dev)
    timeit("rmsnorm_bwd [Me,768] +dres", lambda: ops.rmsnorm_bwd(dy, x, w, rs, dy, dx, dw), bytes_=4.0 * Me * d * ES)
    dh, dyd = rnd(Me, d), torch.empty_like(x)
    timeit("rmsnorm_bwd [Me,768] in place + dropout_bwd(dx) (the step's form)",
           lambda: ops.rmsnorm_bwd(dy, x, w, rs, dh, dh, dw, dx_drop=dyd, drop_out=drop), bytes_=5.0 * Me * d * ES)
    timeit("dropout_apply [Me,768]", lambda: ops.dropout_apply(x, y, drop), bytes_=2.0 * Me * d * ES)

if "misc" in only:
    n = 222_903_552 // 4 * 4
    p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
    v.abs_()
    sh = torch.empty(n, dtype=T, device=dev)
    ns = torch.zeros(1, device=dev)
    timeit("sumsq 223M", lambda: ops.sumsq(g, ns), bytes_=4.0 * n)
    timeit("adamw 223M (+bf16 shadow)", lambda: ops.adamw_step(p, g, m, v, sh, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-6,
                                                               weight_decay=1e-4, gnorm_sq=ns, max_norm=1.0, grad_scale=1.0),
           bytes_=(16 + 12 + ES) * n)
    timeit("zero grads 223M", lambda: g.zero_(), bytes_=4.0 * n)
    src, dst = torch.randn(3072, 768, device=dev), torch.empty(768, 3072, dtype=T, device=dev)
    timeit("transpose_cast 3072x768", lambda: ops.transpose_cast(src, dst), bytes_=3072 * 768 * (4 + ES))

if "index" in only:
    # exact inner-product search of 1024 queries over the reference's 300 600 facts × 256 dims, top-500
    nq, nf, dim, k = 1024, 300600, 256, 500
    q, e = torch.randn(nq, dim, device=dev), torch.randn(nf, dim, device=dev)
    sc = torch.empty(nq, nf, device=dev)
    vals, idx = torch.empty(nq, k, device=dev), torch.empty(nq, k, dtype=torch.int64, device=dev)
    timeit("index scores [1024,256]x[300600,256] fp32", lambda: ops.gemm_nt(q, e, sc), flops=2.0 * nq * nf * dim)
    timeit("index top-500 of [1024,300600]", lambda: ops.topk(sc, k, vals, idx), bytes_=5.0 * nq * nf * 4)
    # the product-quantised variant (faiss.IndexPQ): 16 / 32 / 64 one-byte sub-quantisers over the same facts
    for M in (16, 32, 64):
        ksub, dsub = 256, dim // M
        cent = torch.randn(M, ksub, dsub, device=dev)
        codes = torch.empty(nf, M, dtype=torch.uint8, device=dev)
        sums, counts, err = torch.zeros(M, ksub, dsub, device=dev), torch.zeros(M, ksub, dtype=torch.int32, device=dev), torch.zeros(1, device=dev)
        lut = torch.empty(nq, M, ksub, device=dev)
        timeit(f"pq M={M}: encode 300600 x 256 (nearest of 256 centroids per sub-vector)", lambda: ops.pq_assign(e, cent, codes), flops=3.0 * nf * dim * ksub)
        timeit(f"pq M={M}: one k-means iteration on 65536 vectors (assign + accumulate)", lambda: ops.pq_assign(e[:65536], cent, None, sums, counts, err))
        timeit(f"pq M={M}: lookup tables of 1024 queries", lambda: ops.pq_lut(q, cent, lut))
        timeit(f"pq M={M}: scan [1024 queries] x [300600 codes] ({1024 * 300600 * M / 1e9:.1f} G lookups)", lambda: ops.pq_scan(lut, codes, sc), bytes_=4.0 * nq * nf)

if "retriever" in only:
    # BERT-base bi-encoder forward (src/model.py:451-478) at passage_maxlength 130 — embedding throughput
    from lako_amd.retriever import Retriever, RetrieverConfig
    cfg = RetrieverConfig(apply_passage_mask=True)
    rt = Retriever(cfg, dtype=T).cuda()
    nb, L = 512, 130
    ids = torch.randint(1, cfg.vocab_size, (nb, L), device=dev)
    lens = torch.randint(20, L + 1, (nb,), device=dev)
    mask = torch.arange(L, device=dev)[None, :] < lens[:, None]
    fl = nb * L * cfg.num_hidden_layers * (2.0 * 768 * 768 * 4 + 2.0 * 768 * 3072 * 2 + 4.0 * L * 768) + 2.0 * nb * L * 768 * 256
    timeit(f"retriever embed_text {nb} x {L} BERT-base", lambda: rt.embed_text(ids, mask, "f", True, False), flops=fl)
    ops.reset_timers() if hasattr(ops, "reset_timers") else None
