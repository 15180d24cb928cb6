import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from lako_amd.ops import HipOps
ops = HipOps(); dev = "cuda"
T = torch.bfloat16
H, dk = 4, 64; inner = H*dk
Bn, Lmax = 5, 70; lens = [70, 33, 0, 48, 1]
def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)
off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)
rows = int(off[-1])
km = (torch.arange(Lmax, device=dev)[None] < torch.tensor(lens, device=dev)[:, None]).to(torch.uint8)
rel = rnd(H, 2*Lmax-1, seed=44)
bias = dict(rel_bias=rel, rel_off=Lmax-1)
def pack(tp): return torch.cat([tp[b, :lens[b]] for b in range(Bn)], 0)[None].contiguous()
qkv_p = rnd(Bn, Lmax, 3*inner, dtype=T, seed=47, scale=0.5); qkv_r = pack(qkv_p)
heads = lambda t, c0: t[:, :, c0:c0+inner].unflatten(2, (H, dk))
args_p = tuple(heads(qkv_p, c) for c in (0, inner, 2*inner)); args_r = tuple(heads(qkv_r, c) for c in (0, inner, 2*inner))
rag = dict(q_off=off, k_off=off, max_q=Lmax, max_k=Lmax)
out_p = torch.zeros(Bn, Lmax, inner, dtype=T, device=dev); st_p = torch.zeros(Bn, H, Lmax, 4, device=dev)
ops.attn_fwd(*args_p, out_p.unflatten(2, (H, dk)), st_p, key_mask=km, **bias)
out_r = torch.zeros(1, rows, inner, dtype=T, device=dev); st_r = torch.zeros_like(st_p)
ops.attn_fwd(*args_r, out_r.unflatten(2, (H, dk)), st_r, **bias, **rag)
print("fwd equal", torch.equal(out_r, pack(out_p)))
dout_p = rnd(Bn, Lmax, inner, dtype=T, seed=48) * km[:, :, None].to(T)
drel_p = torch.zeros_like(rel); drel_r = torch.zeros_like(rel)
dqkv_p, dqkv_r = torch.zeros_like(qkv_p), torch.zeros_like(qkv_r)
ops.attn_bwd(*args_p, out_p.unflatten(2, (H, dk)), dout_p.unflatten(2, (H, dk)), st_p, heads(dqkv_p, 0), heads(dqkv_p, inner), heads(dqkv_p, 2*inner), key_mask=km, drel=drel_p, **bias)
ops.attn_bwd(*args_r, out_r.unflatten(2, (H, dk)), pack(dout_p).unflatten(2, (H, dk)), st_r, heads(dqkv_r, 0), heads(dqkv_r, inner), heads(dqkv_r, 2*inner), drel=drel_r, **bias, **rag)
a, b = dqkv_r[0].float(), pack(dqkv_p)[0].float()
d = (a != b)
print("torch.equal", torch.equal(dqkv_r, pack(dqkv_p))); print("mismatch count", int(d.sum()), "of", d.numel())
rowsbad = d.any(1).nonzero().flatten().tolist()
print("bad rows", rowsbad[:20], "offsets", off.tolist())
colsbad = d.any(0).nonzero().flatten()
print("bad col ranges: q", int((colsbad < inner).sum()), "k", int(((colsbad >= inner) & (colsbad < 2*inner)).sum()), "v", int((colsbad >= 2*inner).sum()))
if d.any():
    i = d.nonzero()[0]; print("first", i.tolist(), a[i[0], i[1]].item(), b[i[0], i[1]].item(), "max abs diff", (a-b).abs().max().item())
print("delta stats equal:", torch.equal(st_r[..., 2][km.bool()[:, None, :].expand(-1, H, -1)], st_p[..., 2][km.bool()[:, None, :].expand(-1, H, -1)]))

ai, bi = dqkv_r[0].view(torch.int16), pack(dqkv_p)[0].view(torch.int16)
di = (ai != bi)
print("bit mismatches", int(di.sum()))
if di.any():
    idx = di.nonzero()
    for i in idx[:8]:
        print(i.tolist(), a[i[0], i[1]].item(), b[i[0], i[1]].item(), hex(ai[i[0], i[1]].item() & 0xffff), hex(bi[i[0], i[1]].item() & 0xffff))
# determinism + which one is off
def run_r():
    d = torch.zeros_like(qkv_r); dr = torch.zeros_like(rel)
    ops.attn_bwd(*args_r, out_r.unflatten(2, (H, dk)), pack(dout_p).unflatten(2, (H, dk)), st_r, heads(d, 0), heads(d, inner), heads(d, 2*inner), drel=dr, **bias, **rag)
    return d
def run_p():
    d = torch.zeros_like(qkv_p); dr = torch.zeros_like(rel)
    ops.attn_bwd(*args_p, out_p.unflatten(2, (H, dk)), dout_p.unflatten(2, (H, dk)), st_p, heads(d, 0), heads(d, inner), heads(d, 2*inner), key_mask=km, drel=dr, **bias)
    return pack(d)
r1, r2, p1, p2 = run_r(), run_r(), run_p(), run_p()
print("ragged deterministic", torch.equal(r1, r2), "padded deterministic", torch.equal(p1, p2), "r==p", torch.equal(r1, p1))
os.environ["X"] = "1"
from tests.ref_ops import RefOps
ref = RefOps()
qc, kc, vc = (t.float().cpu() for t in args_p)
dqr, dkr, dvr = (torch.zeros(Bn, Lmax, H, dk) for _ in range(3))
ref.attn_bwd(qc, kc, vc, out_p.unflatten(2, (H, dk)).float().cpu(), dout_p.unflatten(2, (H, dk)).float().cpu(), st_p.cpu(), dqr, dkr, dvr, rel_bias=rel.cpu(), rel_off=Lmax-1, key_mask=km.cpu())
want = dqr[1, 9, 0]
print("ref   ", want[[8, 22, 28, 30]].tolist())
print("ragged", r1[0, 79, [8, 22, 28, 30]].float().tolist())
print("padded", p1[0, 79, [8, 22, 28, 30]].float().tolist())
refp = pack(torch.cat([dqr.reshape(Bn, Lmax, inner), dkr.reshape(Bn, Lmax, inner), dvr.reshape(Bn, Lmax, inner)], 2).to(dev))[0]
for name, t in (("ragged", r1[0].float()), ("padded", p1[0].float())):
    e = (t[:, :inner] - refp[:, :inner]).view(rows, H, dk).norm(dim=2) / (refp[:, :inner].view(rows, H, dk).norm(dim=2) + 1e-9)
    top = torch.topk(e.flatten(), 5)
    print(name, "dq rel err per (row, head): median %.4f max" % e.median().item(), [(int(i) // H, int(i) % H, round(v.item(), 4)) for v, i in zip(top.values, top.indices)])
