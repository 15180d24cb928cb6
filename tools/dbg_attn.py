import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.environ["LAKO_ATTN_PERSIST"] = os.environ.get("LAKO_ATTN_PERSIST", "16")
from lako_amd.ops import HipOps
import ref_ops
from test_kernels_gpu import make_attn, ATTN_CASES, rnd, dev
ops = HipOps(); ref = ref_ops.RefOps()
T = torch.bfloat16
for case in [c for c in ATTN_CASES if c[0] in ("enc_fast", "enc_fast_drop", "enc_fast_odd")]:
    q, k, v, rel, rel_off, km, causal, drop = make_attn(case, T)
    Bn, Lq, H, dk = q.shape; Lk = k.shape[1]
    kw = dict(rel_bias=rel, rel_off=rel_off, key_mask=km, causal=causal, causal_off=0, drop=drop)
    outr = torch.zeros(Bn, Lq, H, dk, device=dev()); st = torch.zeros(Bn, H, Lq, 4, device=dev())
    ref.attn_fwd(q, k, v, outr, st, **kw)
    dout = rnd(Bn, Lq, H, dk, dtype=T, seed=25); o_in = outr.to(T)
    inner = H * dk
    dqkv = torch.zeros(Bn, Lq, 3 * inner, dtype=T, device=dev())
    dq = dqkv[:, :, :inner].view(Bn, Lq, H, dk); dk_ = dqkv[:, :, inner:2*inner].view(Bn, Lk, H, dk); dv = dqkv[:, :, 2*inner:].view(Bn, Lk, H, dk)
    dqr, dkr, dvr = (torch.zeros(t.shape, device=dev()) for t in (q, k, v))
    drel = torch.zeros_like(rel); drelr = torch.zeros_like(rel)
    st2 = st.clone()
    ops.attn_bwd(q, k, v, o_in, dout, st2, dq, dk_, dv, drel=drel, **kw)
    ref.attn_bwd(q, k, v, o_in, dout, st, dqr, dkr, dvr, drel=drelr, **kw)
    for nm, a_, b_ in (("dq", dq, dqr), ("dk", dk_, dkr), ("dv", dv, dvr), ("drel", drel, drelr), ("delta", st2[..., 2], st[..., 2])):
        e = (a_.float() - b_.float()).abs()
        print(case[0], nm, "max err", float(e.max()), "scale", float(b_.abs().max()), "bad frac", float((e > 0.05 * b_.abs().max()).float().mean()))
    # where are dq errors
    e = (dq.float() - dqr).abs().amax(dim=(2, 3))  # [Bn, Lq]
    print("dq err by (b, q//16):", [[round(float(e[b, i*16:(i+1)*16].max()), 2) for i in range((Lq+15)//16)] for b in range(Bn)])
