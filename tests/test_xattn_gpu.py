"""-m gpu: the encoder-state-space cross-attention kernels (csrc/xattn.hip) against their fp32 torch restatements
(tests/ref_ops.py), and the whole re-associated cross-attention — expand, scores, softmax, context, contract, and its
backward incl. the encoder-state gradient — against the reference formulation (project E to K / V, attend;
src/model.py:286-349) evaluated in fp32 on the same bf16 inputs, dropout masks included."""
import numpy as np
import pytest
import torch

from tests.ref_ops import RefOps

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from lako_amd.ops import HipOps
    return HipOps()


@pytest.fixture(scope="module")
def ref():
    return RefOps()


def dev():
    return torch.device("cuda:0")


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def offsets(lens):
    """k_off (packed rows) and p_off (segments padded to 256 columns) of samples with `lens` keys"""
    k = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    p = np.concatenate([[0], np.cumsum([(n + 255) // 256 * 256 for n in lens])]).astype(np.int32)
    return torch.from_numpy(k).to(dev()), torch.from_numpy(p).to(dev()), int(p[-1])


CASES = [  # (lens, T, H, D)
    ([300, 129, 1, 128], 8, 12, 768),      # T5-base rows (R = 96), ragged samples incl. a single key and an exact tile
    ([1000, 777], 8, 16, 1024),            # T5-large rows (R = 128)
    ([200, 90, 513], 1, 12, 768),          # one decoder step of generate (R = 12)
    ([257, 64], 13, 12, 768),              # R = 156 > 128: two row chunks
    ([20000, 14873], 8, 16, 1024),         # BASELINE config 5: T5-large, 100 passages x 200 tokens per sample, full and ragged
]
C5 = CASES[4]


@pytest.mark.parametrize("lens,T,H,D", CASES)
def test_scores_and_context(ops, ref, lens, T, H, D):
    B, R = len(lens), T * H
    k_off, p_off, ptot = offsets(lens)
    E = rnd(sum(lens), D, dtype=BF, seed=1)
    buf = rnd(B, 2 * R + 5, D, dtype=BF, seed=2)           # query rows inside per-sample blocks of a bigger buffer
    Q = buf[:, 3:3 + R]
    S = torch.full((R, ptot + 128), 3.0, device=dev())
    ops.xattn_scores(Q, E, k_off, p_off, ptot, S[:, :ptot])
    Sr = torch.full_like(S, 3.0)
    ref.xattn_scores(Q, E, k_off, p_off, ptot, Sr[:, :ptot])
    scale = Sr[:, :ptot].abs().max().item()
    assert (S - Sr).abs().max().item() <= 2e-5 * scale, ((S - Sr).abs().max().item(), scale)   # fp32 accumulation order only
    assert torch.equal(S[:, ptot:], Sr[:, ptot:])

    P = torch.zeros(R, ptot, dtype=BF, device=dev())
    for b, n in enumerate(lens):
        P[:, int(p_off[b]):int(p_off[b]) + n] = rnd(R, n, dtype=BF, scale=0.05, seed=10 + b).abs()
    for splits in (1, 3, 8):
        full = torch.full((splits, B, R + 2, D + 64), 5.0, device=dev())
        out = full[:, :, :R, :D]                                # slabs inside a bigger buffer: nothing around them is touched
        outr = torch.zeros(splits, B, R, D, device=dev())
        ops.xattn_context(P, E, k_off, p_off, out)
        ref.xattn_context(P, E, k_off, p_off, outr)
        assert rel_l2(out.sum(0), outr.sum(0)) < 1e-5, (splits, rel_l2(out.sum(0), outr.sum(0)))
        assert (out.sum(0) - outr.sum(0)).abs().max().item() <= 1e-4 * outr.abs().max().item()
        assert float((full[:, :, R:] - 5.0).abs().max()) == 0 and float((full[..., D:] - 5.0).abs().max()) == 0
        again = torch.zeros_like(out)
        ops.xattn_context(P, E, k_off, p_off, again)
        assert torch.equal(again, out)                          # no atomics: bit-identical from run to run


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("lens,T,H,D", CASES[:3] + [C5])
def test_softmax_fwd_bwd(ops, ref, lens, T, H, D, p):
    B, R = len(lens), T * H
    k_off, p_off, ptot = offsets(lens)
    max_keys = max(lens) + 7
    drop = (p, 1234, 77) if p > 0 else None
    S = torch.zeros(R, ptot, device=dev())
    dP = torch.zeros(R, ptot, device=dev())
    for b, n in enumerate(lens):
        S[:, int(p_off[b]):int(p_off[b]) + n] = rnd(R, n, scale=3.0, seed=20 + b)
        dP[:, int(p_off[b]):int(p_off[b]) + n] = rnd(R, n, scale=0.5, seed=30 + b)
    st, str_ = torch.zeros(B, R, 2, device=dev()), torch.zeros(B, R, 2, device=dev())
    P = torch.full((R, ptot), 9.0, dtype=BF, device=dev())
    Pr = torch.full((R, ptot), 9.0, dtype=BF, device=dev())
    ops.xattn_softmax_fwd(S, st, P, k_off, p_off, T, H, max_keys, drop)
    ref.xattn_softmax_fwd(S, str_, Pr, k_off, p_off, T, H, max_keys, drop)
    assert torch.allclose(st, str_, rtol=1e-5, atol=1e-6)
    assert torch.equal(P == 0, Pr == 0)                     # the same probabilities are dropped, the padding is zero
    assert (P.float() - Pr.float()).abs().max().item() <= 2 ** -8 * Pr.float().abs().max().item()
    if p > 0:
        kept = [float((P[:, int(p_off[b]):int(p_off[b]) + n] != 0).float().mean()) for b, n in enumerate(lens) if n > 100]
        assert all(abs(k - 0.9) < 0.02 for k in kept), kept
    dS = torch.full((R, ptot), 9.0, dtype=BF, device=dev())
    dSr = torch.full((R, ptot), 9.0, dtype=BF, device=dev())
    ops.xattn_softmax_bwd(S, dP, st, dS, k_off, p_off, T, H, max_keys, drop)
    ref.xattn_softmax_bwd(S, dP, str_, dSr, k_off, p_off, T, H, max_keys, drop)
    assert rel_l2(dS.float(), dSr.float()) < 4e-3, rel_l2(dS.float(), dSr.float())     # bf16 outputs


@pytest.mark.parametrize("Bz,T,H,D", [(16, 8, 12, 768), (3, 5, 16, 1024), (16, 1, 12, 768)])
def test_headbatch(ops, ref, Bz, T, H, D):
    inner = H * 64
    # expand: [B·T, 64] x [64, D] per head, B operand = a column block of a transposed weight, output inside per-sample blocks
    A = rnd(Bz * T, inner, dtype=BF, seed=3).view(Bz, T, H, 64)
    Wt = rnd(D, 5 * inner, dtype=BF, scale=0.05, seed=4)
    Bw = Wt[:, 2 * inner:3 * inner].unflatten(1, (H, 64)).permute(1, 0, 2)          # [H, D, 64]
    big = torch.zeros(Bz, 3 * T * H, D, dtype=BF, device=dev())
    C = big[:, T * H:2 * T * H].unflatten(1, (T, H))
    Cr = torch.zeros(Bz, T, H, D, dtype=BF, device=dev())
    ops.headbatch_nt(A, Bw, C)
    ref.headbatch_nt(A, Bw, Cr)
    assert rel_l2(C.float(), Cr.float()) < 3e-3
    assert float(big[:, :T * H].abs().max()) == 0 and float(big[:, 2 * T * H:].abs().max()) == 0
    # contract: fp32 [B·T·H, D] rows x [64, D] per head
    A2 = rnd(Bz * T * H, D, seed=5).view(Bz, T, H, D)
    W = rnd(5 * inner, D, dtype=BF, scale=0.05, seed=6)
    Bw2 = W[inner:2 * inner].unflatten(0, (H, 64))                                  # [H, 64, D]
    C2 = torch.zeros(Bz * T, inner, dtype=BF, device=dev()).view(Bz, T, H, 64)
    C2r = torch.zeros_like(C2)
    ops.headbatch_nt(A2, Bw2, C2)
    ref.headbatch_nt(A2, Bw2, C2r)
    assert rel_l2(C2.float(), C2r.float()) < 3e-3
    # weight gradient: G[h·64 + j, :] += Σ_m A[m, h·64 + j]·B[m, h, :]
    G = rnd(5 * inner, D, seed=7)
    Gr = G.clone()
    ops.headbatch_tn(A, A2, G[inner:2 * inner].unflatten(0, (H, 64)))
    ref.headbatch_tn(A, A2, Gr[inner:2 * inner].unflatten(0, (H, 64)))
    assert rel_l2(G - rnd(5 * inner, D, seed=7), Gr - rnd(5 * inner, D, seed=7)) < 3e-3
    assert torch.equal(G[:inner], Gr[:inner]) and torch.equal(G[2 * inner:], Gr[2 * inner:])
    # the same product for several problems in ONE launch (lako_headbatch_tn_multi: the deferred Wk / Wv gradients of all decoder
    # layers) equals the single launches bit for bit — 26 problems: more than one launch's 24; fp32 B operands in two slabs too
    probs, single = [], []
    for i in range(26):
        Ai = rnd(Bz * T, inner, dtype=BF, seed=40 + i).view(Bz, T, H, 64)
        Bi = rnd(2, Bz * T * H, D, seed=80 + i).view(2, Bz, T, H, D) if i % 2 else rnd(Bz * T * H, D, seed=80 + i).view(Bz, T, H, D)
        Gi = rnd(inner, D, seed=120 + i)
        probs.append((Ai, Bi, Gi.unflatten(0, (H, 64))))
        Gs = Gi.clone()
        single.append((Ai, Bi, Gs))
    ops.headbatch_tn_multi(probs)
    for Ai, Bi, Gs in single:
        ops.headbatch_tn(Ai, Bi, Gs.unflatten(0, (H, 64)))
    for (_, _, Gm), (_, _, Gs) in zip(probs, single):
        assert torch.equal(Gm.flatten(0, 1), Gs)


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("lens,T,H,D,tol", [([333, 128, 50], 8, 12, 768, 1.5e-2), (C5[0], 8, 16, 1024, 1e-2)], ids=["base_333", "c5_20000"])
def test_reassociated_cross_attention_equals_projected(ops, ref, p, lens, T, H, D, tol):
    """The whole chain against the reference formulation in fp32: K = E·Wkᵀ, V = E·Wvᵀ, softmax(q·Kᵀ)·V and its autograd
    (src/model.py:286-349) — at T5-base rows with short ragged samples, and at BASELINE config 5's size: T5-large rows (16 heads,
    d = 1024) over 20 000 and 14 873 keys per sample, the kernels the config-5 bench line runs."""
    B, R, inner = len(lens), T * H, H * 64
    k_off, p_off, ptot = offsets(lens)
    max_keys = max(lens) + 67
    drop = (p, 99, 5) if p > 0 else None
    E = rnd(sum(lens), D, dtype=BF, seed=1)
    q = rnd(B * T, inner, dtype=BF, scale=0.3, seed=2)
    W = rnd(2 * inner, D, dtype=BF, scale=0.06, seed=3)       # rows: Wk | Wv
    Wt = W.t().contiguous()
    dctx = rnd(B * T, inner, dtype=BF, scale=0.2, seed=4)

    # ---- reference: projected keys / values, fp32 autograd, the attention dropout recipe at bh = b·H + h, q = t, k = s
    from tests.ref_ops import attn_keep_mask, drop_key
    Ef = E.float().requires_grad_(True)
    qf = q.float().requires_grad_(True)
    Wf = W.float().requires_grad_(True)
    ctx_ref = torch.zeros(B, T, H, 64, device=dev())
    outs = []
    for b, n in enumerate(lens):
        Eb = Ef[int(k_off[b]):int(k_off[b]) + n]
        K = (Eb @ Wf[:inner].T).view(n, H, 64)
        V = (Eb @ Wf[inner:].T).view(n, H, 64)
        s = torch.einsum("thd,shd->hts", qf.view(B, T, H, 64)[b], K)
        pn = torch.softmax(s, -1)
        if p > 0:
            keep = attn_keep_mask((b + 1) * H, T, max_keys, drop_key(99, 5), p, dev())[b * H:, :, :n]
            pn = torch.where(keep, pn / (1 - np.float32(p)), torch.zeros_like(pn))
        outs.append(torch.einsum("hts,shd->thd", pn, V))
    ctx_ref = torch.stack(outs)                                # [B, T, H, 64]
    gE, gq, gW = torch.autograd.grad(ctx_ref, [Ef, qf, Wf], dctx.float().view(B, T, H, 64))

    # ---- the kernels (one decoder layer: rows [dC' | Q'] of the per-sample blocks)
    DQ = torch.zeros(B, 2 * R, D, dtype=BF, device=dev())
    Qp = DQ[:, R:]
    ops.headbatch_nt(q.view(B, T, H, 64), Wt[:, :inner].unflatten(1, (H, 64)).permute(1, 0, 2), Qp.unflatten(1, (T, H)))
    S = torch.zeros(R, ptot, device=dev())
    ops.xattn_scores(Qp, E, k_off, p_off, ptot, S)
    PS = torch.zeros(2 * R, ptot, dtype=BF, device=dev())
    st = torch.zeros(B, R, 2, device=dev())
    ops.xattn_softmax_fwd(S, st, PS[:R], k_off, p_off, T, H, max_keys, drop)
    Cp = torch.zeros(2, B, R, D, device=dev())
    ops.xattn_context(PS[:R], E, k_off, p_off, Cp)
    ctx = torch.zeros(B * T, inner, dtype=BF, device=dev())
    ops.headbatch_nt(Cp.unflatten(2, (T, H)), W[inner:].unflatten(0, (H, 64)), ctx.view(B, T, H, 64))
    errs = {"ctx": rel_l2(ctx.float().view(B, T, H, 64), ctx_ref)}
    assert errs["ctx"] < 1e-2, errs        # (measured at 20 000 keys: ctx 0.005, dq / dW / dE 0.005-0.006)
    # backward
    G = torch.zeros(2 * inner, D, device=dev())
    dCp = DQ[:, :R]
    ops.headbatch_nt(dctx.view(B, T, H, 64), Wt[:, inner:].unflatten(1, (H, 64)).permute(1, 0, 2), dCp.unflatten(1, (T, H)))
    ops.headbatch_tn(dctx.view(B, T, H, 64), Cp.unflatten(2, (T, H)), G[inner:].unflatten(0, (H, 64)))
    dP = torch.zeros(R, ptot, device=dev())
    ops.xattn_scores(dCp, E, k_off, p_off, ptot, dP)
    ops.xattn_softmax_bwd(S, dP, st, PS[R:], k_off, p_off, T, H, max_keys, drop)
    dQp = torch.zeros(3, B, R, D, device=dev())
    ops.xattn_context(PS[R:], E, k_off, p_off, dQp)
    dq = torch.zeros(B * T, inner, dtype=BF, device=dev())
    ops.headbatch_nt(dQp.unflatten(2, (T, H)), W[:inner].unflatten(0, (H, 64)), dq.view(B, T, H, 64))
    ops.headbatch_tn(q.view(B, T, H, 64), dQp.unflatten(2, (T, H)), G[:inner].unflatten(0, (H, 64)))
    dE = torch.zeros(sum(lens) + 8, D, device=dev())
    items = []
    for b, n in enumerate(lens):
        n8 = (n + 7) // 8 * 8                                   # the rows past n add the zero padding columns of PS
        items.append((PS[:, int(p_off[b]):int(p_off[b]) + n8], DQ[b], dE[int(k_off[b]):int(k_off[b]) + n8], 1.0))
    ops.gemm_tn_grouped(items, split_k=1)
    errs.update(dq=rel_l2(dq.float(), gq), dW=rel_l2(G, gW), dE=rel_l2(dE[:sum(lens)], gE))
    import json, os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"parity_xattn_chain_{max(lens)}keys_p{p}.json"), "w") as f:
            json.dump(errs, f)
    assert errs["dq"] < tol and errs["dW"] < tol and errs["dE"] < tol, errs
    assert float(dE[sum(lens):].abs().max()) == 0


@pytest.mark.parametrize("lens,H,D,Z", [([300, 129, 1, 33], 12, 768, 4), ([1000, 777], 16, 1024, 16), ([64, 5000, 31], 8, 512, 7)])
def test_decode_step_one_pass(ops, ref, lens, H, D, Z):
    """lako_xattn_decode + lako_xattn_decode_combine (one decode step: scores, softmax and context in one pass over the encoder
    states, key ranges merged afterwards) against the fp32 restatement and against softmax(Q′·Eᵀ)·E·Wvᵀ computed directly."""
    B = len(lens)
    k_off, _, _ = offsets(lens)
    E = rnd(sum(lens), D, dtype=BF, seed=1)
    Q = rnd(B, H + 3, D, dtype=BF, scale=0.08, seed=2)[:, 1:1 + H]
    Wv = rnd(H * 64, D, dtype=BF, scale=0.05, seed=3)
    pml, pc = torch.zeros(Z, B, 16, 2, device=dev()), torch.zeros(Z, B, 16, D, device=dev())
    pml_r, pc_r = torch.zeros_like(pml), torch.zeros_like(pc)
    ops.xattn_decode(Q, E, k_off, pml, pc)
    ref.xattn_decode(Q, E, k_off, pml_r, pc_r)
    assert torch.equal(torch.isinf(pml[:, :, :H, 0]), torch.isinf(pml_r[:, :, :H, 0]))
    fin = ~torch.isinf(pml_r[:, :, :H, 0])
    assert torch.allclose(pml[:, :, :H][fin], pml_r[:, :, :H][fin], rtol=2e-3, atol=1e-4)       # Σ exp of bf16-rounded … no: fp32 exps
    ctx, ctx_r = torch.zeros(B, H * 64, dtype=BF, device=dev()), torch.zeros(B, H * 64, dtype=BF, device=dev())
    ops.xattn_decode_combine(pml, pc, Wv, ctx, H)
    ref.xattn_decode_combine(pml_r, pc_r, Wv, ctx_r, H)
    assert rel_l2(ctx.float(), ctx_r.float()) < 6e-3, rel_l2(ctx.float(), ctx_r.float())
    direct = torch.zeros(B, H, 64, device=dev())
    for b, n in enumerate(lens):
        e = E[int(k_off[b]):int(k_off[b]) + n].float()
        p = torch.softmax(Q[b].float() @ e.T, -1)
        direct[b] = torch.einsum("hc,hjc->hj", p @ e, Wv.float().view(H, 64, D))
    assert rel_l2(ctx.float().view(B, H, 64), direct) < 1e-2, rel_l2(ctx.float().view(B, H, 64), direct)
