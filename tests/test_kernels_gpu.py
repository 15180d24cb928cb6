"""-m gpu: every HIP kernel behind the C-ABI against the fp32 torch restatement of its contract
(tests/ref_ops.py), on the same seeded inputs.  fp32 kernels (exact-f32 MFMA) must agree to ~1e-4;
bf16 kernels are compared against the reference evaluated on the same bf16-rounded inputs with a
tolerance of a few bf16 ulps of the output scale (stated per test)."""
import numpy as np
import pytest
import torch

from tests.ref_ops import RefOps, keep_mask

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}


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
    g = torch.Generator(device="cpu").manual_seed(seed + int(np.prod(shape)) % 9973)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def close(got, want, dtype, what, k=1.0, tight=False):
    """|got − want| <= atol + rtol·|want|, with `want` the fp32 test double's UNROUNDED result.  Two bound sets:
    tight=True — for kernels that do fp32 arithmetic on the given operands and round ONCE (every GEMM / weight-gradient / norm / loss /
      optimizer kernel): the bound follows the OUTPUT dtype and the accumulation, not the input dtype (round 4; the round-3 bounds accepted
      a bf16-in / fp32-out GEMM at 1.5 % of the output scale):
        fp32 out of fp32 operands        2e-5·scale + 2e-4·|want|   (unchanged)
        fp32 out of bf16 operands        1e-4·scale + 1e-4·|want|   (exact products, fp32 accumulation: only the summation order differs)
        bf16 out                         3e-5·scale + 4.2e-3·|want| (one rounding to bf16: half an ulp is 2^-8 = 3.9e-3 of the value)
    tight=False — kernels with bf16 INTERMEDIATES (attention: probabilities and score gradients are rounded to bf16 before the second
      product): 1.5e-2·scale + 2e-2·|want| for bf16, the fp32 bounds for fp32.
    k scales both terms (sums over many rows: k > 1)."""
    out_bf16 = got.dtype == torch.bfloat16
    got, want = got.float(), want.float()
    scale = max(want.abs().max().item(), 1e-6)
    if dtype == torch.float32:
        atol, rtol = 2e-5 * scale * k + 1e-6, 2e-4 * k
    elif not tight:
        atol, rtol = 1.5e-2 * scale * k, 2e-2 * k
    elif out_bf16:
        atol, rtol = 3e-5 * scale * k + 1e-6, 4.2e-3 * max(k, 1.0)
    else:
        atol, rtol = 1e-4 * scale * k + 1e-6, 1e-4 * k
    err = (got - want).abs()
    bad = err > atol + rtol * want.abs()
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.3e} "
                                 f"(scale {scale:.3e}, atol {atol:.2e} rtol {rtol:.1e}) first bad idx {bad.nonzero()[0].tolist()}")
    assert torch.isfinite(got).all(), f"{what}: non-finite output"


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 264, 200), (108, 64, 32), (1024, 768, 768), (77, 2304, 64),
                                   (256, 128, 3072)])
def test_gemm_nt_plain(ops, ref, dt, M, N, K):
    T = DT[dt]
    A, B = rnd(M, K, dtype=T, seed=1), rnd(N, K, dtype=T, seed=2)
    for out_t in (T, torch.float32):
        C = torch.full((M, N), 7.0, dtype=out_t, device=dev())
        Cr = torch.zeros(M, N, device=dev())
        ops.gemm_nt(A, B, C, alpha=0.5)
        ref.gemm_nt(A, B, Cr, alpha=0.5)
        close(C, Cr, T, f"gemm_nt {dt}->{out_t} {M}x{N}x{K}", tight=True)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7, 8, 9])
@pytest.mark.parametrize("M,N,K", [(1024, 768, 768), (700, 520, 200), (256, 256, 32), (2048, 2304, 768), (300, 264, 3072), (128, 768, 72),
                                   (128, 768, 768), (100, 200, 160), (16, 3072, 768), (130, 776, 3072)])
def test_gemm_nt_tile_variants(ops, ref, variant, M, N, K):
    """every tile variant of the bf16 NT kernel (0: 128², 1: 256×128, 2: 256² 2-buffer, 3 / 9 (round 6): the four-wave kernels with the
    hand-placed K loop, 192- / 256-row tiles — shapes and epilogues they do not take run on variant 2, 4: the 128²
    4-slot ring that skinny problems are dispatched to, 5: the 64² kernel whose four waves split K, for M <= 256 rows — it
    falls back to the ring when K is not a multiple of 32, 7: 192×256 tiles — the 256² kernel's MT = 6 instantiation, whose last
    wave keeps its epilogue scratch behind the K-slice buffers, 8: 288×256 tiles — MT = 9: the last A piece of a K-slice exists for
    waves 0–3 only, the last epilogue pass is 16 rows; plain epilogues only, the others fall back to 256-row tiles),
    persistent and one-tile-per-workgroup grids, ragged edges, fused epilogues."""
    T = torch.bfloat16
    A, B = rnd(M, K, dtype=T, seed=41), rnd(N, K, dtype=T, seed=42)
    R = rnd(M, N, dtype=T, seed=43)
    try:
        # (persistent grid, LDS-transposed wide epilogue, banded tile order forced with 3 tile-rows per band)
        for persistent, wide, group_m in ((1, 1, 8), (0, 0, 0), (1, 1, -3), (0, 1, -2)):
            ops.set_tuning("gemm_nt_variant", variant)
            ops.set_tuning("gemm_nt_persistent", persistent)
            ops.set_tuning("gemm_nt_wide_epi", wide)
            ops.set_tuning("gemm_nt_group_m", group_m)
            for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(aux=R, aux_scale=1.1)):
                C = torch.empty(M, N, dtype=T, device=dev())
                Cr = torch.zeros(M, N, device=dev())
                ops.gemm_nt(A, B, C, **kw)
                ref.gemm_nt(A, B, Cr, **kw)
                close(C, Cr, T, f"gemm_nt variant {variant} persistent {persistent} wide {wide} group_m {group_m} {list(kw)} {M}x{N}x{K}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_variant", -1)
        ops.set_tuning("gemm_nt_persistent", 1)
        ops.set_tuning("gemm_nt_wide_epi", 1)
        ops.set_tuning("gemm_nt_group_m", 8)


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (700, 520, 200), (515, 264, 72), (256, 524, 64)])
def test_gemm_nt_side_operand_in_lds(ops, ref, M, N, K):
    """256² bf16 tiles with a residual or an aux-mask operand: the LDS-staged operand + row-major stores epilogue must give
    the generic epilogue's result BIT FOR BIT (same arithmetic, in the accumulator layout), on ragged edges, with row strides
    wider than the row, and fall back when a row is not 16-byte granular (N = 524)."""
    T = torch.bfloat16
    A, B = rnd(M, K, dtype=T, seed=51), rnd(N, K, dtype=T, seed=52)
    Rw, Xw = rnd(M, N + 24, dtype=T, seed=53), rnd(M, N + 8, dtype=T, seed=54)
    R, X = Rw[:, :N], Xw[:, 8:]
    Cw = torch.zeros(M, N + 16, dtype=T, device=dev())
    cases = (dict(resid=R), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(resid=R, relu=True), dict(aux=X, aux_scale=1.1),
             dict(aux=X, aux_scale=1.0 / 0.9, drop=(0.1, 3, 4)))
    try:
        for variant in (2, 7, 8):           # 256-row tiles (4 side passes per wave), 192-row tiles (3) and 288-row tiles (9 passes of 16 rows)
            ops.set_tuning("gemm_nt_variant", variant)
            for kw in cases:
                got = {}
                for side in (1, 0):
                    ops.set_tuning("gemm_nt_side_lds", side)
                    Cw.fill_(7.0)
                    C = Cw[:, 8:8 + N]
                    ops.gemm_nt(A, B, C, **kw)
                    assert torch.all(Cw[:, :8] == 7.0) and torch.all(Cw[:, 8 + N:] == 7.0), "stores outside the output columns"
                    got[side] = C.clone()
                assert torch.equal(got[0], got[1]), f"side-in-LDS epilogue differs from the generic one {list(kw)} {M}x{N}x{K} variant {variant}"
                Cr = torch.zeros(M, N, device=dev())
                ref.gemm_nt(A, B, Cr, **kw)
                close(got[1], Cr, T, f"gemm_nt side operand {list(kw)} {M}x{N}x{K} variant {variant}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_variant", -1)
        ops.set_tuning("gemm_nt_side_lds", 1)


@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (700, 520, 200), (515, 264, 64), (256, 256, 128), (16700, 1024, 192), (33300, 520, 72),
                                   (47757, 768, 3072), (8192, 2304, 768)])
def test_gemm_nt_eight_phase_equals_two_phase(ops, ref, M, N, K):
    """`gemm_nt_pp` = 1 (round 4, the default): the 8-phase main loop of the 256² kernel — quadrant-wise fragment reads, half-tile LDS-DMA
    stagings 1¾ K-steps ahead retired by counted waits, waves 4–7 one barrier behind waves 0–3.  Every accumulator still sums its K-halves
    in the same order, so the result must equal the two-phase loop's BIT FOR BIT — two K-steps (K = 72: the second almost empty; K = 128),
    three, many; K = 64 falls back to the two-phase loop; ragged edges; persistent workgroups that walk several tiles (264 … 560 tiles on
    256 workgroups: the stream of stagings crosses tile boundaries, the epilogue's scratch is the buffer the next tile's second K-step
    is staged into afterwards); plain, wide, and LDS-staged side-operand epilogues.  Repeated: a mis-placed wait or read shows up as a
    rare wrong tile, not as a steady failure."""
    T = torch.bfloat16
    A, B = rnd(M, K, dtype=T, seed=71), rnd(N, K, dtype=T, seed=72)
    R = rnd(M, N, dtype=T, seed=73)
    cases = (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(aux=R, aux_scale=1.1))
    try:
        ops.set_tuning("gemm_nt_variant", 2)
        for wide in (1, 0):
            ops.set_tuning("gemm_nt_wide_epi", wide)
            for kw in cases:
                ops.set_tuning("gemm_nt_pp", 0)
                ops.set_tuning("gemm_nt_glds", 0)
                want = torch.full((M, N), 7.0, dtype=T, device=dev())
                ops.gemm_nt(A, B, want, **kw)
                for pp, glds in ((1, 1), (1, 0), (0, 1)):       # (global_load_lds staging: rows past the edge are clamped, not zero-filled)
                    ops.set_tuning("gemm_nt_pp", pp)
                    ops.set_tuning("gemm_nt_glds", glds)
                    for rep in range(3 if M * N * K < 4e10 else 2):
                        C = torch.full((M, N), 7.0, dtype=T, device=dev())
                        ops.gemm_nt(A, B, C, **kw)
                        bad = (C != want).any(dim=1).nonzero().flatten()
                        assert bad.numel() == 0, (f"pp {pp} glds {glds} differs from the two-phase / buffer-load kernel {list(kw)} wide {wide} "
                                                  f"{M}x{N}x{K} rep {rep}: {bad.numel()} rows, first {bad[:8].tolist()}")
                if M * N <= 2048 * 768:
                    Cr = torch.zeros(M, N, device=dev())
                    ref.gemm_nt(A, B, Cr, **kw)
                    close(want, Cr, T, f"gemm_nt {list(kw)} {M}x{N}x{K}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_variant", -1)
        ops.set_tuning("gemm_nt_wide_epi", 1)
        ops.set_tuning("gemm_nt_pp", 0)
        ops.set_tuning("gemm_nt_glds", 1)


@pytest.mark.parametrize("M,N,K", [(128, 768, 768), (128, 768, 3072), (100, 200, 160), (256, 2304, 2304), (16, 3072, 4096), (130, 776, 1056)])
def test_gemm_nt_skinny_tiles(ops, ref, M, N, K):
    """the decoder's GEMM kernel (K split over the workgroup's waves) with both tile sizes and every fused epilogue, bf16 and
    fp32 output"""
    T = torch.bfloat16
    A, B = rnd(M, K, dtype=T, seed=61), rnd(N, K, dtype=T, seed=62)
    R, Rf = rnd(M, N, dtype=T, seed=63), rnd(M, N, seed=64)
    try:
        ops.set_tuning("gemm_nt_variant", 5)
        for tiles in (2, 3):
            ops.set_tuning("gemm_nt_skinny", tiles)
            for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(aux=R, aux_scale=1.1)):
                C = torch.empty(M, N, dtype=T, device=dev())
                Cr = torch.zeros(M, N, device=dev())
                ops.gemm_nt(A, B, C, **kw)
                ref.gemm_nt(A, B, Cr, **kw)
                close(C, Cr, T, f"gemm_nt skinny tiles {tiles} {list(kw)} {M}x{N}x{K}", tight=True)
            for kw in (dict(alpha=0.25), dict(resid=Rf)):
                C = torch.empty(M, N, device=dev())
                Cr = torch.zeros(M, N, device=dev())
                ops.gemm_nt(A, B, C, **kw)
                ref.gemm_nt(A, B, Cr, **kw)
                close(C, Cr, T, f"gemm_nt skinny f32 out tiles {tiles} {list(kw)} {M}x{N}x{K}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_variant", -1)
        ops.set_tuning("gemm_nt_skinny", 1)


def test_gemm_nt_tail_split(ops, ref):
    """270 tiles of 256² = one full round of the persistent grid + 15 tiles: the rows of the full round go to the 256²
    kernel, the rest to a second launch with small tiles; dropout draws must use the rows' GLOBAL index."""
    T = torch.bfloat16
    M, N, K = 256 * 90, 768, 160
    A, B = rnd(M, K, dtype=T, seed=81), rnd(N, K, dtype=T, seed=82)
    R, X = rnd(M, N, dtype=T, seed=83), rnd(M, N, dtype=T, seed=84)
    try:
        for split in (1, 0):
            ops.set_tuning("gemm_nt_tail_split", split)
            for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(aux=X, aux_scale=1.1)):
                C = torch.empty(M, N, dtype=T, device=dev())
                Cr = torch.zeros(M, N, device=dev())
                ops.gemm_nt(A, B, C, **kw)
                ref.gemm_nt(A, B, Cr, **kw)
                close(C, Cr, T, f"gemm_nt tail split {split} {list(kw)}", tight=True)
                if "drop" in kw:
                    assert torch.equal(C == 0, Cr.to(T) == 0) or (C == 0).float().mean().item() > 0.05
    finally:
        ops.set_tuning("gemm_nt_tail_split", 1)


def test_gemm_nt_tile_queue(ops, ref):
    """`gemm_nt_queue` (data-parallel overlap mode): the persistent 256² kernel pulls every tile after a workgroup's first from
    per-XCD ticket counters.  Every tile must be computed exactly once — results bit-identical to the strided walk — with every
    epilogue, on multi-round launches (banded and row-major tile order), back to back (the last workgroup leaves the counters zero
    for the next launch), on a second stream (its own counters), and on shapes where the queue does not apply (one round; K of
    one K-step) and the launch falls back to the strided kernel."""
    T = torch.bfloat16
    try:
        for (M, N, K) in [(70000, 768, 768), (40000, 2304, 192), (5000, 768, 64), (1000, 768, 768)]:
            A, B = rnd(M, K, dtype=T, seed=95), rnd(N, K, dtype=T, seed=96)
            R, X = rnd(M, N, dtype=T, seed=97), rnd(M, N, dtype=T, seed=98)
            for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(aux=X, aux_scale=1.1)):
                got = []
                for q in (0, 1, 1):
                    ops.set_tuning("gemm_nt_queue", q)
                    C = torch.empty(M, N, dtype=T, device=dev())
                    ops.gemm_nt(A, B, C, **kw)
                    got.append(C)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    C2 = torch.empty(M, N, dtype=T, device=dev())
                    ops.gemm_nt(A, B, C2, **kw)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2]) and torch.equal(got[0], C2), (M, N, K, list(kw))
            Cr = torch.zeros(M, N, device=dev())
            ref.gemm_nt(A, B, Cr)
            ops.set_tuning("gemm_nt_queue", 1)
            C = torch.empty(M, N, dtype=T, device=dev())
            ops.gemm_nt(A, B, C)
            close(C, Cr, T, f"gemm_nt tile queue {M}x{N}x{K}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_queue", 0)


@pytest.mark.parametrize("M,N,K", [(128, 2304, 768), (128, 3072, 768), (16, 768, 768), (200, 1024, 1024), (37, 776, 512), (128, 768, 2048),
                                   (300, 768, 768)])
def test_gemm_nt_norm_fused(ops, ref, M, N, K):
    """LAKO_EPI_NORM_A (round 4): the product on the T5-RMSNorm of A's rows, formed inside the M <= 256 kernel — against the two
    launches it replaces (lako_rmsnorm_fwd + the plain product): the normalised rows and rstd handed back are what rmsnorm_fwd writes
    (rstd to fp32 summation order, rows to one bf16 rounding), the product agrees with the reference, with fused epilogues; shapes the
    kernel does not take (K > 1024, rows > 256) run as two launches through the same call."""
    T = torch.bfloat16
    A, B = rnd(M, K, dtype=T, seed=71) * 3.0, rnd(N, K, dtype=T, seed=72)
    w = (1.0 + 0.2 * rnd(K, seed=73)).contiguous()
    R = rnd(M, N, dtype=T, seed=74)
    eps = 1e-6
    xn0, rs0 = torch.empty(M, K, dtype=T, device=dev()), torch.empty(M, device=dev())
    ops.rmsnorm_fwd(A, w, xn0, rs0, eps)
    for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5)):
        C0 = torch.empty(M, N, dtype=T, device=dev())
        ops.gemm_nt(xn0, B, C0, **kw)
        C1 = torch.empty(M, N, dtype=T, device=dev())
        xn1, rs1 = torch.full((M, K), 7.0, dtype=T, device=dev()), torch.full((M,), 7.0, device=dev())
        ops.gemm_nt(A, B, C1, norm=(w, eps, xn1, rs1), **kw)
        torch.testing.assert_close(rs1, rs0, rtol=2e-6, atol=0)
        assert float((xn1.float() - xn0.float()).abs().max()) <= 2.0 ** -7 * float(xn0.float().abs().max())      # one bf16 ulp at most
        assert float((xn1 != xn0).float().mean()) < 1e-3
        Cr = torch.zeros(M, N, device=dev())
        ref.gemm_nt(xn1, B, Cr, **kw)                    # the product of the rows the kernel says it used
        close(C1, Cr, T, f"gemm_nt norm-fused {list(kw)} {M}x{N}x{K}", tight=True)
        close(C1, C0.float(), T, f"gemm_nt norm-fused vs two launches {list(kw)} {M}x{N}x{K}")


@pytest.mark.parametrize("knobs", [dict(gemm_nt_queue=1), dict(gemm_nt_tile288=0), dict()], ids=["queue", "no288", "default"])
def test_gemm_nt_norm_refused_before_any_launch(ops, knobs):
    """(round 5, ADVICE) LAKO_EPI_NORM_A on a shape only the big-tile kernels take — M = 5 600 rows x N = 3 072: 22 x 12 = 264 tiles of 256²,
    i.e. a tail plan with the 256-row tiles (tile queue on, or 288-row tiles off) — must be refused with LAKO_E_UNSUPPORTED before ANYTHING is
    launched: round 4 ran the head rows of the tail plan on the un-normalised A before refusing the flag.  C, the normalised rows and
    rstd stay untouched; the Python op then runs the two launches."""
    T = torch.bfloat16
    M, N, K = 5600, 3072, 768
    A, B = rnd(M, K, dtype=T, seed=75) * 3.0, rnd(N, K, dtype=T, seed=76)
    w = (1.0 + 0.2 * rnd(K, seed=77)).contiguous()
    try:
        for k_, v_ in knobs.items():
            ops.set_tuning(k_, v_)
        C1 = torch.full((M, N), 7.0, dtype=T, device=dev())
        xn1, rs1 = torch.full((M, K), 7.0, dtype=T, device=dev()), torch.full((M,), 7.0, device=dev())
        rc = ops._gemm_nt_call(A, B, C1, 1.0, False, None, None, 1.0, None, False, (w, 1e-6, xn1, rs1))
        torch.cuda.synchronize()
        assert rc == -4, rc                                     # LAKO_E_UNSUPPORTED
        assert bool((C1 == 7.0).all()) and bool((xn1 == 7.0).all()) and bool((rs1 == 7.0).all()), "a refused call must not launch anything"
        # the op's fallback: norm, then the plain product — equal to doing the two by hand
        ops.gemm_nt(A, B, C1, norm=(w, 1e-6, xn1, rs1))
        xn0, rs0 = torch.empty_like(xn1), torch.empty_like(rs1)
        ops.rmsnorm_fwd(A, w, xn0, rs0, 1e-6)
        C0 = torch.empty_like(C1)
        ops.gemm_nt(xn0, B, C0)
        assert torch.equal(C1, C0) and torch.equal(xn1, xn0) and torch.equal(rs1, rs0)
    finally:
        ops.set_tuning("gemm_nt_queue", 0)
        ops.set_tuning("gemm_nt_tile288", 1)


def test_gemm_nt_tile_height_plan(ops, ref):
    """(`gemm_nt_tile288`, round 4, on by default: 288-row tiles where they save the tail launch or a round — two rounds of 498 tiles here.)
    With `gemm_nt_tile192` on, launch_nt prices 256-row and 192-row tiles per call: 47 757 rows x 768 columns — the benchmark's
    encoder shape — take 192-row tiles in three full rounds and no tail launch; with it off (the default: measured no faster,
    profiles/r03e_gemm_tile192.txt) the call runs 256-row tiles + the tail launch.  Same result to bf16 rounding of identical fp32 sums (bit for bit: each output element's K-loop is the
    same sequence of MFMAs), with every fused epilogue; dropout draws by global row."""
    T = torch.bfloat16
    M, N, K = 47757, 768, 96
    A, B = rnd(M, K, dtype=T, seed=91), rnd(N, K, dtype=T, seed=92)
    R, X = rnd(M, N, dtype=T, seed=93), rnd(M, N, dtype=T, seed=94)
    try:
        for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5), dict(aux=X, aux_scale=1.1)):
            got = []
            for t192, t288 in ((1, 0), (0, 0), (0, 1)):      # (0, 1) = the default: 288-row tiles for the plain epilogues of this shape
                ops.set_tuning("gemm_nt_tile192", t192)
                ops.set_tuning("gemm_nt_tile288", t288)
                ops.probe = []
                C = torch.empty(M, N, dtype=T, device=dev())
                ops.gemm_nt(A, B, C, **kw)
                torch.cuda.synchronize()
                ops.probe = None
                got.append(C)
            assert torch.equal(got[0], got[1]) and torch.equal(got[2], got[1]), list(kw)
            Cr = torch.zeros(M, N, device=dev())
            ref.gemm_nt(A, B, Cr, **kw)
            close(got[0], Cr, T, f"gemm_nt 192-row plan {list(kw)}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_tile192", 0)
        ops.set_tuning("gemm_nt_tile288", 1)
        ops.probe = None


@pytest.mark.parametrize("M,N,K", [(47757, 768, 256), (16500, 1032, 384), (65600, 264, 128 * 5), (9000, 2304, 768), (25000, 520, 3072)])
def test_gemm_nt_four_wave_kernels(ops, ref, M, N, K):
    """(round 6, csrc/gemm_nt4.h) The four-wave kernels with the hand-placed K loop — two K-slices of LDS-DMA in flight, counted waits, the
    stream running on into the workgroup's next tile — forced (variant 9: 256-row tiles, 3: 192-row tiles) and as the default plan
    (`gemm_nt_four`, which picks the height by the rounds of the chip), against the eight-wave kernels of rounds 1-5 (`gemm_nt_four` 0):
    BIT-identical outputs (each element's K loop is the same sequence of MFMAs, the epilogue the same fp32 operations in the same order and
    one rounding), for every epilogue the four-wave kernels take and — through the fallback — for those they do not (ReLU + residual);
    ragged last tiles in M and N, several tiles per workgroup (the DMA stream crosses tile boundaries), an odd number of K-slice pairs."""
    T = torch.bfloat16
    A, B = rnd(M, K, dtype=T, seed=71), rnd(N, K, dtype=T, seed=72)
    R, X = rnd(M, N, dtype=T, seed=73), rnd(M, N, dtype=T, seed=74)
    try:
        for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(drop=(0.1, 9, 10)), dict(alpha=0.5, drop=(0.1, 5, 6)), dict(resid=R, drop=(0.1, 7, 8), alpha=0.5),
                   dict(resid=R), dict(aux=X, aux_scale=1.1), dict(relu=True, resid=R)):
            got = {}
            # (gemm_nt_four 3: the dropout epilogues through the LDS transposition like the others; 1: straight from the accumulator layout)
            for name, four, variant in (("eight-wave", 0, -1), ("default plan", 1, -1), ("256-row", 1, 9), ("192-row", 1, 3), ("256-row, LDS epilogues", 3, 9),
                                        ("192-row, LDS epilogues", 3, 3)):
                ops.set_tuning("gemm_nt_four", four)
                ops.set_tuning("gemm_nt_variant", variant)
                C = torch.full((M, N), float("nan"), dtype=T, device=dev())
                ops.gemm_nt(A, B, C, **kw)
                torch.cuda.synchronize()
                got[name] = C
            for name in got:
                assert torch.equal(got[name].view(torch.int16), got["eight-wave"].view(torch.int16)), f"{name} vs eight-wave kernels, {list(kw)} {M}x{N}x{K}"
            Cr = torch.zeros(M, N, device=dev())
            ref.gemm_nt(A, B, Cr, **kw)
            close(got["default plan"], Cr, T, f"gemm_nt four-wave {list(kw)} {M}x{N}x{K}", tight=True)
    finally:
        ops.set_tuning("gemm_nt_four", 1)
        ops.set_tuning("gemm_nt_variant", -1)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_gemm_tn_grouped(ops, ref, dt):
    """several weight gradients sharing K in one launch (256×256 kernel, bf16) or its per-item fallback (fp32, small
    shapes); accumulates into C like lako_gemm_tn."""
    T = DT[dt]
    K = 3000
    for shapes in ([(768, 2304), (768, 768), (3072, 768), (768, 3072)], [(256, 264), (520, 256)], [(256, 256), (64, 304)],
                   [(304, 256)] * 9):
        probs, want = [], []
        for i, (M, N) in enumerate(shapes):
            A, B = rnd(K, M, dtype=T, seed=70 + i) * 0.25, rnd(K, N, dtype=T, seed=90 + i) * 0.25
            C0 = rnd(M, N, seed=110 + i)
            Cr = C0.clone()
            ref.gemm_tn(A, B, Cr, alpha=0.5 + i)
            probs.append((A, B, C0, 0.5 + i))
            want.append(Cr)
        ops.gemm_tn_grouped(probs)
        for (A, B, Cg, _), Cr, (M, N) in zip(probs, want, shapes):
            close(Cg, Cr, T, f"gemm_tn_grouped {dt} {M}x{N}", tight=True)


def test_gemm_tn_grouped_slab_reduction(ops, ref):
    """lako_gemm_tn_grouped with a workspace (round 4): the K-splits of a tile meet through fp32 partial tiles + one arrival ticket per
    tile instead of float atomics; the workgroup that arrives last sums the slabs in split order.  Same result as the atomic path within
    fp32 summation noise, BIT-IDENTICAL from run to run (atomics were not), C accumulated (not overwritten), 2 / 3 / 4 splits, a K
    that is not a multiple of 64, ragged tile edges; the scratch may be dirty (it is reused across launches)."""
    T = torch.bfloat16
    scratch = {}

    def ws(n):
        if scratch.get("t") is None or scratch["t"].numel() < n:
            scratch["t"] = torch.full((n,), 0x7F, dtype=torch.uint8, device=dev())      # garbage on purpose
        return scratch["t"]
    for K, split in ((47757, 0), (6400, 2), (9999, 3), (16384, 4)):
        dy = {n: rnd(K, n, dtype=T, seed=11 + n) for n in (768, 2304, 3072, 520)}
        x = {n: rnd(K, n, dtype=T, seed=13 + n) for n in (768, 3072, 264)}
        shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768), (520, 264)]
        runs = []
        for rep in range(3):
            G = [torch.full(sh, 0.25, device=dev()) for sh in shapes]
            items = [(dy[m], x[n], g, 0.5) for (m, n), g in zip(shapes, G)]
            ops.gemm_tn_grouped(items, split_k=split, workspace=ws if rep < 2 else None)
            runs.append(G)
        Gr = [torch.full(sh, 0.25, device=dev()) for sh in shapes]
        ref.gemm_tn_grouped([(dy[m], x[n], g, 0.5) for (m, n), g in zip(shapes, Gr)])
        for a, b, c, r, sh in zip(runs[0], runs[1], runs[2], Gr, shapes):
            assert torch.equal(a, b), f"slab reduction not reproducible K={K} split={split} {sh}"
            close(a, r, T, f"gemm_tn_grouped slabs K={K} split={split} {sh}", tight=True)
            close(c, r, T, f"gemm_tn_grouped atomics K={K} split={split} {sh}", tight=True)
    assert scratch["t"] is not None


def test_gemm_tn_exclusive_and_overwrite(ops, ref):
    """lako_gemm_tn_grouped with split_k −1 (one contributor, nothing else adds meanwhile: C += v by plain loads / stores) and −2 (C = v:
    no zeroed C), and rows_out < M: a problem whose true row count is not a multiple of 8 touches only its own rows — two such problems
    written side by side into one buffer (the encoder-state gradient of the cross-attention: one problem per sample) never overwrite
    each other's rows, whatever order the workgroups finish in."""
    T = torch.bfloat16
    K = 2304
    ks = [1003, 517, 2049]                       # "keys" per sample, none a multiple of 8
    tot = sum(ks)
    P = rnd(K, 3 * 2304, dtype=T, seed=81)
    D = [rnd(K, 768, dtype=T, seed=82 + i) for i in range(3)]
    off = [0, ks[0], ks[0] + ks[1]]
    for mode in (-2, -1):
        out = torch.full((tot + 8, 768), 3.0, device=dev())
        want = out.clone()
        items, items_r = [], []
        p0 = 0
        for i, nk in enumerate(ks):
            n8 = -(-nk // 8) * 8
            A = P[:, p0:p0 + n8].clone()
            A[:, nk:] = 0                          # (the score matrices' zero padding columns)
            items.append((A, D[i], out[off[i]:off[i] + n8], 0.5, nk))
            items_r.append((A, D[i], want[off[i]:off[i] + n8], 0.5, nk))
            p0 += 2304
        ops.gemm_tn_grouped(items, split_k=mode)
        ref.gemm_tn_grouped(items_r, split_k=mode)
        close(out, want, T, f"gemm_tn_grouped split_k {mode}", tight=True)
        assert torch.all(out[tot:] == 3.0)
    # one problem, K = 128 (the decoder's deferred weight gradients): exclusive read-modify-write accumulates
    A, B = rnd(128, 768, dtype=T, seed=91), rnd(128, 3072, dtype=T, seed=92)
    C = torch.full((768, 3072), 1.0, device=dev())
    Cr = C.clone()
    ops.gemm_tn_grouped([(A, B, C, 1.0)] * 1 + [(B, A, torch.zeros(3072, 768, device=dev()), 1.0)], split_k=-1)
    ref.gemm_tn_grouped([(A, B, Cr, 1.0)], split_k=-1)
    close(C, Cr, T, "gemm_tn_grouped exclusive rmw", tight=True)
    # an output whose rows are not 16-byte aligned (row stride 3074 floats) takes the dword form of the same epilogue
    for mode in (-1, -2):
        big = torch.full((768, 3074), 2.0, device=dev())
        Cv, want = big[:, 1:3073], torch.full((768, 3072), 2.0, device=dev())
        ops.gemm_tn_grouped([(A, B, Cv, 0.5), (B, A, torch.zeros(3072, 768, device=dev()), 1.0)], split_k=mode)
        ref.gemm_tn_grouped([(A, B, want, 0.5)], split_k=mode)
        close(Cv, want, T, f"gemm_tn_grouped split_k {mode}, unaligned rows", tight=True)
        assert torch.all(big[:, 0] == 2.0) and torch.all(big[:, 3073] == 2.0)


@pytest.mark.parametrize("extra", [0, 8, 136])
def test_gemm_tn_grouped_hybrid_schedule(ops, ref, extra):
    """lako_gemm_tn_grouped split_k −3 (round 4): whole rounds of 256 tiles run their WHOLE K in one workgroup each (plain adds), the
    remaining `extra` tiles are cut into K-pieces that add by atomics — C += alpha·Aᵀ·B for every problem, onto non-zero C, with a K
    that is not a multiple of 64 (the full-K units stage its incomplete step first) and ragged tile edges."""
    T = torch.bfloat16
    K = 4813
    shapes = [(768, 3072)] * 7 + [(768, 768)] * 0 + [(256, 256)] * 4            # 252 + 4 = 256 tiles
    shapes += {0: [], 8: [(520, 1000)], 136: [(768, 3072)] * 3 + [(1792, 1024)]}[extra]      # + 3 × 4 = 12 → the kernel sees cdiv tiles: 8 ↔ (520,1000) = 3 × 4 = 12
    probs, probs_r = [], []
    for i, (M, N) in enumerate(shapes):
        A, B = rnd(K, M, dtype=T, seed=300 + i) * 0.25, rnd(K, N, dtype=T, seed=400 + i) * 0.25
        C = rnd(M, N, seed=500 + i)
        probs.append((A, B, C, 0.5))
        probs_r.append((A, B, C.clone(), 0.5))
    ops.gemm_tn_grouped(probs, split_k=-3)
    ref.gemm_tn_grouped(probs_r, split_k=-3)
    for (_, _, C, _), (_, _, Cr, _), sh in zip(probs, probs_r, shapes):
        close(C, Cr, T, f"gemm_tn_grouped hybrid {sh} extra {extra}")


@pytest.mark.parametrize("K,M,N", [(1024, 256, 256), (1000, 512, 768), (200, 256, 520), (4813, 768, 3072), (47757, 768, 768), (131, 264, 256)])
def test_gemm_tn_four_wave_kernel(ops, ref, K, M, N):
    """(round 6, csrc/gemm_tn4.h) The 256 x 256 weight-gradient kernel on four waves with the hand-placed K loop (`gemm_tn_four`, default)
    against the eight-wave kernel of rounds 1-5: with ONE contributor per output element (split_k -1 read-modify-write onto non-zero C,
    -2 overwrite) the two are BIT-identical (same MFMAs in the same order: the incomplete K-step first, then the whole ones); with K-splits
    the float atomics add in a different order (compared with the fp32 reference).  K ranges with and without an incomplete step, odd and
    even step counts, ragged tile edges, a K too short for the kernel (131 rows: falls back)."""
    T = torch.bfloat16
    A, B = rnd(K, M, dtype=T, seed=81) * 0.25, rnd(K, N, dtype=T, seed=82) * 0.25
    C0 = rnd(M, N, seed=83)
    try:
        for split in (-1, -2, 0, 3):
            got = []
            for four in (0, 1):
                ops.set_tuning("gemm_tn_four", four)
                C = C0.clone()
                ops.gemm_tn(A, B, C, alpha=0.5, split_k=split)
                torch.cuda.synchronize()
                got.append(C)
            if split < 0:
                assert torch.equal(got[0], got[1]), f"four-wave vs eight-wave weight-gradient kernel, split_k {split}, K {K} M {M} N {N}"
            Cr = C0.clone()
            ref.gemm_tn(A, B, Cr, alpha=0.5, split_k=split)
            close(got[1], Cr, T, f"gemm_tn four-wave split_k {split} K {K} M {M} N {N}", tight=True)
    finally:
        ops.set_tuning("gemm_tn_four", 1)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 768, 32128), (40, 264, 5000), (128, 132, 2048)])
def test_gemm_nt_split_k_atomic(ops, ref, dt, M, N, K):
    """few tiles × very long K with fp32 atomic accumulation: the ring kernel splits K over workgroups (LM-head backward)."""
    T = DT[dt]
    A, B = rnd(M, K, dtype=T, seed=61) * 0.25, rnd(N, K, dtype=T, seed=62) * 0.25
    C = rnd(M, N, seed=63)
    Cr = C.clone()
    ops.gemm_nt(A, B, C, alpha=0.5, atomic=True)
    ref.gemm_nt(A, B, Cr, alpha=0.5, atomic=True)
    close(C, Cr, T, f"gemm_nt split-K atomic {dt} {M}x{N}x{K}", tight=True)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_gemm_nt_epilogues(ops, ref, dt):
    T = DT[dt]
    M, N, K = 200, 136, 96
    A, B = rnd(M, K, dtype=T, seed=3), rnd(N, K, dtype=T, seed=4)
    R = rnd(M, N, dtype=T, seed=5)
    aux = rnd(M, N, dtype=T, seed=6)
    drop = (0.1, 1234, 77)
    cases = [dict(relu=True, drop=drop), dict(resid=R, drop=drop), dict(aux=aux, aux_scale=1.0 / 0.9),
             dict(relu=True), dict(resid=R)]
    for kw in cases:
        C = torch.empty(M, N, dtype=T, device=dev())
        Cr = torch.zeros(M, N, device=dev())
        ops.gemm_nt(A, B, C, **kw)
        ref.gemm_nt(A, B, Cr, **kw)
        close(C, Cr, T, f"gemm_nt epi {list(kw)} {dt}", tight=True)
    # strided views: A = middle slice of a wider matrix, C = slice of a wider output, atomic accumulate
    wide = rnd(M, 3 * K, dtype=T, seed=7)
    Cw = torch.ones(M, 2 * N, dtype=torch.float32, device=dev())
    Cwr = Cw.clone()
    ops.gemm_nt(wide[:, K:2 * K], B, Cw[:, N:], atomic=True)
    ref.gemm_nt(wide[:, K:2 * K], B, Cwr[:, N:], atomic=True)
    close(Cw, Cwr, T, f"gemm_nt strided/atomic {dt}", tight=True)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("K,M,N,split", [(500, 264, 136, 0), (108, 64, 32, 1), (4096, 768, 768, 0), (130, 128, 128, 3),
                                         (64, 96, 2304, 0), (1000, 520, 264, 0), (777, 256, 768, 5),
                                         (6400, 2304, 768, 0)])
def test_gemm_tn(ops, ref, dt, K, M, N, split):
    T = DT[dt]
    A, B = rnd(K, M, dtype=T, seed=8), rnd(K, N, dtype=T, seed=9)
    C = torch.ones(M, N, device=dev())
    Cr = C.clone()
    ops.gemm_tn(A, B, C, alpha=0.25, split_k=split)
    ref.gemm_tn(A, B, Cr, alpha=0.25)
    close(C, Cr, T, f"gemm_tn {dt} K{K} {M}x{N}", tight=True)
    # strided operands (column slices of wider activations)
    Aw, Bw = rnd(K, 2 * M, dtype=T, seed=10), rnd(K, 3 * N, dtype=T, seed=11)
    C.fill_(0)
    Cr.fill_(0)
    ops.gemm_tn(Aw[:, M:], Bw[:, N:2 * N], C)
    ref.gemm_tn(Aw[:, M:], Bw[:, N:2 * N], Cr)
    close(C, Cr, T, f"gemm_tn strided {dt}", tight=True)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("rows,d", [(37, 32), (1000, 768), (1001, 768), (1, 768), (513, 1024), (64, 512)])
def test_rmsnorm(ops, ref, dt, rows, d):
    T = DT[dt]
    x = rnd(rows, d, dtype=T, seed=12)
    w = 1.0 + 0.1 * rnd(d, seed=13)
    for drop in (None, (0.1, 5, 9)):
        y, yr = torch.empty_like(x), torch.empty(rows, d, device=dev())
        rs, rsr = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
        ops.rmsnorm_fwd(x, w, y, rs, 1e-6, drop)
        ref.rmsnorm_fwd(x, w, yr, rsr, 1e-6, drop)
        close(rs, rsr, torch.float32, "rstd", tight=True)
        close(y, yr, T, f"rmsnorm_fwd {dt} drop={drop}", tight=True)
        dy, dres = rnd(rows, d, dtype=T, seed=14), rnd(rows, d, dtype=T, seed=15)
        for dr in (None, dres):
            dx, dxr = torch.empty_like(x), torch.empty(rows, d, device=dev())
            dw, dwr = torch.ones(d, device=dev()), torch.ones(d, device=dev())
            ops.rmsnorm_bwd(dy, x, w, rsr, dr, dx, dw, drop)
            ref.rmsnorm_bwd(dy, x, w, rsr, dr, dxr, dwr, drop)
            close(dx, dxr, T, f"rmsnorm_bwd dx {dt}", tight=True)
            close(dw, dwr, torch.float32, f"rmsnorm_bwd dw {dt}", k=20)
            # second output: dropout_bwd(dx) for the next residual branch — bit-identical to a separate lako_dropout_apply
            dx2, dd, dw2 = torch.empty_like(x), torch.empty_like(x), torch.ones(d, device=dev())
            ops.rmsnorm_bwd(dy, x, w, rsr, dr, dx2, dw2, drop, dx_drop=dd, drop_out=(0.2, 11, 12))
            want = torch.empty_like(x)
            ops.dropout_apply(dx, want, (0.2, 11, 12))
            assert torch.equal(dx2, dx) and torch.equal(dd, want)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_embed_dropout(ops, ref, dt):
    T = DT[dt]
    V, d, n = 96, 64, 300
    table = rnd(V, d, dtype=T, seed=16)
    ids = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(1)).to(dev())
    for drop in (None, (0.25, 3, 4)):
        out, outr = torch.empty(n, d, dtype=T, device=dev()), torch.empty(n, d, device=dev())
        ops.embed_fwd(ids, table, out, drop)
        ref.embed_fwd(ids, table, outr, drop)
        assert torch.equal(out.float(), outr.to(T).float()), "embed_fwd must be exact"
        dout = rnd(n, d, dtype=T, seed=17)
        dt_, dtr = torch.zeros(V, d, device=dev()), torch.zeros(V, d, device=dev())
        ops.embed_bwd(ids, dout, dt_, drop)
        ref.embed_bwd(ids, dout, dtr, drop)
        close(dt_, dtr, torch.float32, "embed_bwd", k=5)
        # the ordered form (LAKO_DETERMINISTIC=1): the same scatter, rows of one id added in position order by one wave — an exact
        # repeat, and on top of a non-zero gradient (the decoder's and the encoder's scatter land in one table)
        perm = torch.argsort(ids, stable=True)
        seen = []
        for _ in range(2):
            dto = torch.full((V, d), 0.5, device=dev())
            ops.embed_bwd_ordered(ids, perm, dout, dto, drop)
            seen.append(dto)
        assert torch.equal(seen[0], seen[1])
        close(seen[0] - 0.5, dtr, torch.float32, "embed_bwd_ordered", k=5)
    x = rnd(4096 * 8, dtype=T, seed=18)
    y, yr = torch.empty_like(x), torch.empty(x.numel(), device=dev())
    ops.dropout_apply(x, y, (0.1, 42, 7))
    ref.dropout_apply(x, yr, (0.1, 42, 7))
    assert torch.equal(y == 0, yr == 0), "dropout keep pattern must be bit-identical to the integer recipe"
    frac = (y == 0).float().mean().item()
    assert abs(frac - 0.1) < 0.01, frac
    close(y, yr, T, "dropout_apply", tight=True)


def test_relpos(ops, ref):
    H, nb, R = 12, 32, 399
    table = rnd(nb, H, seed=19)
    lut = torch.randint(0, nb, (R,), generator=torch.Generator().manual_seed(2)).int().to(dev())
    rel, relr = torch.empty(H, R, device=dev()), torch.empty(H, R, device=dev())
    ops.relpos_expand(table, lut, rel)
    ref.relpos_expand(table, lut, relr)
    assert torch.equal(rel, relr)
    drel = rnd(H, R, seed=20)
    dtab, dtabr = torch.zeros(nb, H, device=dev()), torch.zeros(nb, H, device=dev())
    ops.relpos_reduce(drel, lut, dtab)
    ref.relpos_reduce(drel, lut, dtabr)
    close(dtab, dtabr, torch.float32, "relpos_reduce", k=5)


# ------------------------------------------------------------------------------------------------
ATTN_CASES = [
    # name, Bn, H, Lq, Lk, dk, bias, mask, causal, drop, packed
    ("enc_tiny", 3, 2, 12, 12, 32, True, True, False, None, True),
    ("enc_odd", 2, 4, 37, 37, 32, True, True, False, None, True),
    ("enc_base", 2, 12, 200, 200, 64, True, True, False, None, True),
    ("enc_base_drop", 2, 3, 200, 200, 64, True, True, False, (0.1, 11, 3), True),
    # no key mask, d_head 64, <= 256 keys: the bf16 runs of these go through the fast path (csrc/attn_enc.hip) in the backward
    ("enc_fast", 2, 12, 200, 200, 64, True, False, False, None, True),
    ("enc_fast_drop", 3, 3, 200, 200, 64, True, False, False, (0.1, 17, 6), True),
    ("enc_fast_odd", 2, 4, 37, 37, 64, True, False, False, (0.1, 18, 7), True),
    ("enc_fast_256", 1, 2, 256, 256, 64, True, False, False, None, True),
    ("enc_fast_nobias", 2, 2, 130, 130, 64, False, False, False, None, True),
    ("dec_self", 4, 12, 7, 7, 64, True, False, True, None, True),
    ("dec_self_drop", 4, 2, 9, 9, 32, True, False, True, (0.1, 12, 4), False),
    ("cross", 3, 4, 5, 600, 64, False, True, False, None, False),
    ("cross_drop", 2, 2, 20, 300, 32, False, True, False, (0.1, 13, 5), False),
    ("decode_step", 4, 8, 1, 23, 64, True, False, False, None, False),
]


def make_attn(case, T):
    name, Bn, H, Lq, Lk, dk, bias, mask, causal, drop, packed = case
    inner = H * dk
    if packed and Lq == Lk:
        qkv = rnd(Bn, Lq, 3 * inner, dtype=T, seed=21, scale=0.5)
        q = qkv[:, :, 0:inner].view(Bn, Lq, H, dk)
        k = qkv[:, :, inner:2 * inner].view(Bn, Lk, H, dk)
        v = qkv[:, :, 2 * inner:].view(Bn, Lk, H, dk)
    else:
        q = rnd(Bn, Lq, H, dk, dtype=T, seed=22, scale=0.5)
        kv = rnd(Bn, Lk, 2 * inner, dtype=T, seed=23, scale=0.5)
        k = kv[:, :, :inner].view(Bn, Lk, H, dk)
        v = kv[:, :, inner:].view(Bn, Lk, H, dk)
    rel, rel_off = None, 0
    if bias:
        R = Lq + Lk - 1 if name != "decode_step" else 2 * 50 - 1
        rel = rnd(H, R, seed=24)
        rel_off = Lq - 1 if name != "decode_step" else 49 - (Lk - 1)   # query sits at position Lk-1
    km = None
    if mask:
        g = torch.Generator().manual_seed(3)
        lens = torch.randint(max(1, Lk // 2), Lk + 1, (Bn,), generator=g)
        km = (torch.arange(Lk)[None, :] < lens[:, None])
        km[Bn - 1] = False                      # one fully padded row
        km = km.to(torch.uint8).to(dev())
    return q, k, v, rel, rel_off, km, causal, drop


@pytest.mark.parametrize("dt", ["f32", "bf16", "bf16_persistent", "bf16_fused", "bf16_fused_ring3"])
@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(ops, ref, dt, case, monkeypatch):
    if dt == "bf16_persistent":      # the encoder fast path's persistent kernels (backward here; forward: test_attention_fast_path_forward)
        if not case[0].startswith("enc_fast"):
            pytest.skip("persistent kernels serve the encoder fast path only")
        monkeypatch.setenv("LAKO_ATTN_PERSIST", "15")
        dt = "bf16"
    elif dt.startswith("bf16_fused"):  # round 5: the one-pass backward (enc_bwd_fused_kernel), ring of 4 (default) and of 3 stages
        if not case[0].startswith("enc_fast"):
            pytest.skip("the one-pass backward serves the encoder fast path only")
        monkeypatch.setenv("LAKO_ATTN_PERSIST", "16")
        monkeypatch.setenv("LAKO_ATTN_FUSED_NST", "3" if dt.endswith("ring3") else "4")
        dt = "bf16"
    else:
        monkeypatch.setenv("LAKO_ATTN_PERSIST", "0")
    T = DT[dt]
    q, k, v, rel, rel_off, km, causal, drop = make_attn(case, T)
    Bn, Lq, H, dk = q.shape
    Lk = k.shape[1]
    kw = dict(rel_bias=rel, rel_off=rel_off, key_mask=km, causal=causal, causal_off=0, drop=drop)
    out = torch.zeros(Bn, Lq, H, dk, dtype=T, device=dev())
    outr = torch.zeros(Bn, Lq, H, dk, device=dev())
    st, stg = torch.zeros(Bn, H, Lq, 4, device=dev()), torch.zeros(Bn, H, Lq, 4, device=dev())
    sc, scr = torch.zeros(Bn, H, Lq, Lk, device=dev()), torch.zeros(Bn, H, Lq, Lk, device=dev())
    ops.attn_fwd(q, k, v, out, stg, scores_out=sc, **kw)
    ref.attn_fwd(q, k, v, outr, st, scores_out=scr, **kw)
    close(sc, scr, T, f"attn scores {case[0]} {dt}")
    close(stg[..., 0], st[..., 0], T, f"attn rowmax {case[0]} {dt}")
    close(stg[..., 1], st[..., 1], T, f"attn 1/rowsum {case[0]} {dt}", k=2)
    close(out, outr, T, f"attn_fwd out {case[0]} {dt}")
    # backward (feed the reference's forward products so the comparison isolates the backward kernels)
    dout = rnd(Bn, Lq, H, dk, dtype=T, seed=25)
    o_in = outr.to(T)
    dq, dk_, dv = (torch.zeros_like(t) for t in (q, k, v))
    dq = torch.zeros(q.shape, dtype=T, device=dev()) if not q.is_contiguous() else dq
    # gradient tensors must share the layout of their forward tensors: rebuild the packed layout
    if not q.is_contiguous():
        inner = H * dk
        dqkv = torch.zeros(Bn, Lq, 3 * inner, dtype=T, device=dev())
        dq = dqkv[:, :, :inner].view(Bn, Lq, H, dk)
        dk_ = dqkv[:, :, inner:2 * inner].view(Bn, Lk, H, dk)
        dv = dqkv[:, :, 2 * inner:].view(Bn, Lk, H, dk)
    elif not k.is_contiguous():
        inner = H * dk
        dkv = torch.zeros(Bn, Lk, 2 * inner, dtype=T, device=dev())
        dk_ = dkv[:, :, :inner].view(Bn, Lk, H, dk)
        dv = dkv[:, :, inner:].view(Bn, Lk, H, dk)
    dqr, dkr, dvr = (torch.zeros(t.shape, device=dev()) for t in (q, k, v))
    drel = torch.zeros_like(rel) if rel is not None else None
    drelr = torch.zeros_like(rel) if rel is not None else None
    bkw = dict(rel_bias=rel, rel_off=rel_off, key_mask=km, causal=causal, causal_off=0, drop=drop)
    ops.attn_bwd(q, k, v, o_in, dout, st, dq, dk_, dv, drel=drel, **bkw)
    ref.attn_bwd(q, k, v, o_in, dout, st, dqr, dkr, dvr, drel=drelr, **bkw)
    close(dq, dqr, T, f"attn_bwd dq {case[0]} {dt}", k=2)
    close(dk_, dkr, T, f"attn_bwd dk {case[0]} {dt}", k=2)
    close(dv, dvr, T, f"attn_bwd dv {case[0]} {dt}", k=2)
    if rel is not None:
        close(drel, drelr, T, f"attn_bwd drel {case[0]} {dt}", k=4)


@pytest.mark.parametrize("persist", ["0", "15"], ids=["per_item", "persistent"])
@pytest.mark.parametrize("case", [c for c in ATTN_CASES if c[0].startswith("enc_fast")], ids=lambda c: c[0])
def test_attention_fast_path_forward(ops, ref, case, persist, monkeypatch):
    """Forward of the fast path (bf16, no score capture — test_attention's forward captures scores and therefore runs the generic
    kernel): outputs and softmax statistics against the fp32 double, dropout included (same integer recipe) — the kernel with one
    workgroup per (sequence, head) and the persistent kernel (LDS-DMA prefetch of the next item's K / V images)."""
    monkeypatch.setenv("LAKO_ATTN_PERSIST", persist)
    T = torch.bfloat16
    q, k, v, rel, rel_off, km, causal, drop = make_attn(case, T)
    Bn, Lq, H, dk = q.shape
    kw = dict(rel_bias=rel, rel_off=rel_off, key_mask=km, causal=causal, causal_off=0, drop=drop)
    out = torch.zeros(Bn, Lq, H, dk, dtype=T, device=dev())
    outr = torch.zeros(Bn, Lq, H, dk, device=dev())
    st, stg = torch.zeros(Bn, H, Lq, 4, device=dev()), torch.zeros(Bn, H, Lq, 4, device=dev())
    ops.attn_fwd(q, k, v, out, stg, **kw)
    ref.attn_fwd(q, k, v, outr, st, **kw)
    close(stg[..., 0], st[..., 0], T, f"fast rowmax {case[0]}")
    close(stg[..., 1], st[..., 1], T, f"fast 1/rowsum {case[0]}", k=2)
    close(out, outr, T, f"fast attn_fwd out {case[0]}")


@pytest.mark.parametrize("drop", [None, (0.1, 21, 8)], ids=["nodrop", "drop"])
def test_attention_persistent_kernels_equal_per_item_kernels(ops, drop, monkeypatch):
    """The persistent forward (one 16-wave workgroup per CU walking the sequences of a head, next item's images by LDS-DMA) runs the
    same arithmetic per query block as the kernel with one workgroup per (sequence, head): bit-identical outputs and statistics —
    on ragged packed sequences incl. empty ones, one-token ones, more sequences than a workgroup's first pass (several items per
    workgroup, both image pairs in use), 13 heads (a head count that does not divide the CU count)."""
    T = torch.bfloat16
    H, dk, Lmax = 13, 64, 200
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(1, Lmax + 1, (70,), generator=g).tolist()
    lens[3], lens[17], lens[40], lens[69] = 0, 1, 200, 16
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    rows = int(off[-1])
    inner = H * dk
    qkv = rnd(1, rows, 3 * inner, dtype=T, seed=51, scale=0.5)
    q, k, v = (qkv[:, :, i * inner:(i + 1) * inner].view(1, rows, H, dk) for i in range(3))
    rel = rnd(H, 2 * Lmax - 1, seed=52)
    res = []
    for persist in ("0", "15"):
        monkeypatch.setenv("LAKO_ATTN_PERSIST", persist)
        out = torch.zeros(1, rows, H, dk, dtype=T, device=dev())
        st = torch.zeros(len(lens), H, Lmax, 4, device=dev())
        ops.attn_fwd(q, k, v, out, st, rel_bias=rel, rel_off=Lmax - 1, drop=drop, q_off=off, k_off=off, max_q=Lmax, max_k=Lmax)
        torch.cuda.synchronize()
        res.append((out, st))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1][..., :2], res[1][1][..., :2])
    assert float(res[1][0].float().abs().max()) > 0
    # backward: the persistent dQ and dK/dV passes against the per-item kernels — dq / dk / dv to one bf16 ulp, δ (handed from the
    # dQ to the dK/dV pass through the statistics) and the bias gradient to fp32 summation order
    dout = rnd(1, rows, H, dk, dtype=T, seed=53)
    outs = []
    for persist in ("0", "15", "16"):       # per-item kernels, persistent two-pass kernels, the one-pass kernel (round 5)
        monkeypatch.setenv("LAKO_ATTN_PERSIST", persist)
        st = res[0][1].clone()
        dqkv = torch.zeros(1, rows, 3 * inner, dtype=T, device=dev())
        dq, dk_, dv = (dqkv[:, :, i * inner:(i + 1) * inner].view(1, rows, H, dk) for i in range(3))
        drel = torch.zeros_like(rel)
        ops.attn_bwd(q, k, v, res[0][0], dout, st, dq, dk_, dv, rel_bias=rel, drel=drel, rel_off=Lmax - 1, drop=drop, q_off=off,
                     k_off=off, max_q=Lmax, max_k=Lmax)
        torch.cuda.synchronize()
        outs.append((dqkv, st, drel))
    # (measured: 121 of 7 000 dq rows differ by ONE bf16 ulp — the compiler contracts p·dP′ into different fused multiply-adds in the
    #  two kernels; the forward above is bit-identical)
    for other in outs[1:]:
        for i, nm in enumerate(("dq", "dk", "dv")):
            a0, a1 = (o[0][0, :, i * inner:(i + 1) * inner].float() for o in (outs[0], other))
            assert float((a0 - a1).abs().max()) <= 2.0 ** -7 * max(1.0, float(a0.abs().max())), (nm, float((a0 - a1).abs().max()))
            assert float((a0 != a1).float().mean()) < 0.01, nm
        assert torch.allclose(outs[0][1][..., :3], other[1][..., :3], rtol=1e-5, atol=1e-5)     # (δ: 64 products summed in another order)
        close(other[2], outs[0][2], torch.float32, "persistent drel", k=5)
        assert float(other[0].float().abs().max()) > 0 and float(other[2].abs().max()) > 0


@pytest.mark.parametrize("lengths", ["config2", "short", "full", "mixed"])
def test_attention_fast_path_race_screen(ops, lengths, monkeypatch):
    """The encoder fast path's kernels hand data between waves through LDS rings filled by LDS-DMA across item boundaries, ordered only by
    counted `s_waitcnt vmcnt` + barriers (the one-pass backward: a 4-stage slab ring, a K image staged one item ahead, a dS slab read one
    slab later).  A misplaced wait shows as RARE wrong tiles that come and go with timing — so: the same launch 40 times at the
    benchmark's scale (320 passages x 12 heads = 15 items per workgroup) and at length mixes that move every boundary (1 … 40 tokens:
    an item per slab; all 200; a mix with empty and one-token passages), forward and backward, every output BIT-identical to the first
    run's (the kernels have no atomics on these outputs; the bias gradient's cross-workgroup float atomics are compared to summation
    noise)."""
    monkeypatch.setenv("LAKO_ATTN_PERSIST", "18")
    T = torch.bfloat16
    H, dk, Lmax = 12, 64, 200
    inner = H * dk
    g = torch.Generator().manual_seed(11)
    if lengths == "config2":
        lens = torch.randint(100, 201, (320,), generator=g).tolist()
    elif lengths == "short":
        lens = torch.randint(1, 41, (700,), generator=g).tolist()
    elif lengths == "full":
        lens = [200] * 160
    else:
        lens = torch.randint(1, 201, (300,), generator=g).tolist()
        for i in (0, 17, 18, 150, 299):
            lens[i] = 0
        for i in (5, 6, 151):
            lens[i] = 1
    Bn = len(lens)
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    rows = int(off[-1])
    qkv = rnd(1, rows, 3 * inner, dtype=T, seed=61, scale=0.5)
    dout = rnd(1, rows, inner, dtype=T, seed=62, scale=0.5)
    rel = rnd(H, 2 * Lmax - 1, seed=63)
    heads = lambda t, c0: t[:, :, c0:c0 + inner].unflatten(2, (H, dk))      # noqa: E731
    args = tuple(heads(qkv, c) for c in (0, inner, 2 * inner))
    order = torch.argsort(torch.tensor(lens), descending=True, stable=True).to(torch.int32).to(dev())
    kw = dict(rel_bias=rel, rel_off=Lmax - 1, drop=(0.1, 5, 6), q_off=off, k_off=off, max_q=Lmax, max_k=Lmax, order=order)
    first = None
    for rep in range(40):
        out = torch.zeros(1, rows, inner, dtype=T, device=dev())
        st = torch.zeros(Bn, H, Lmax, 4, device=dev())
        ops.attn_fwd(*args, out.unflatten(2, (H, dk)), st, **kw)
        dqkv = torch.zeros_like(qkv)
        drel = torch.zeros_like(rel)
        ops.attn_bwd(*args, out.unflatten(2, (H, dk)), heads(dout, 0), st, *(heads(dqkv, c) for c in (0, inner, 2 * inner)), drel=drel, **kw)
        torch.cuda.synchronize()
        if first is None:
            first = (out, dqkv, drel)
            assert float(dqkv.float().abs().max()) > 0 and bool(torch.isfinite(dqkv.float()).all())
            continue
        assert torch.equal(out, first[0]), f"forward differs in run {rep}"
        if not torch.equal(dqkv, first[1]):
            bad = (dqkv != first[1]).nonzero()
            raise AssertionError(f"backward differs in run {rep}: {bad.shape[0]} elements, first at {bad[0].tolist()}")
        close(drel, first[2], torch.float32, f"race screen drel run {rep}", k=5)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("persist", ["18", "2", "7", "0"])
def test_attention_processing_order_does_not_change_results(ops, persist, monkeypatch):
    """lako_attn_*_t.order (round 4): the order in which the encoder fast path's workgroups take the sequences — longest first for load
    balance — is a pure scheduling choice: forward outputs, statistics, dq / dk / dv are BIT-identical for the natural order, the
    descending-length order and a random permutation; the bias gradient (float atomics) agrees to summation noise.  Default kernels
    (persistent dQ pass), all-persistent, and all per-item kernels (the non-persistent dQ kernel walks consecutive sequences and ignores
    the order)."""
    monkeypatch.setenv("LAKO_ATTN_PERSIST", persist)
    T = torch.bfloat16
    H, dk, Lmax = 12, 64, 200
    inner = H * dk
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(1, Lmax + 1, (45,), generator=g).tolist()
    lens[7] = 0
    Bn = len(lens)
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    rows = int(off[-1])
    qkv = rnd(1, rows, 3 * inner, dtype=T, seed=48, scale=0.5)
    dout = rnd(1, rows, inner, dtype=T, seed=49, scale=0.5)
    rel = rnd(H, 2 * Lmax - 1, seed=50)
    heads = lambda t, c0: t[:, :, c0:c0 + inner].unflatten(2, (H, dk))      # noqa: E731
    args = tuple(heads(qkv, c) for c in (0, inner, 2 * inner))
    kw = dict(rel_bias=rel, rel_off=Lmax - 1, drop=(0.1, 3, 4), q_off=off, k_off=off, max_q=Lmax, max_k=Lmax)
    orders = [None, torch.argsort(torch.tensor(lens), descending=True, stable=True).to(torch.int32).to(dev()),
              torch.randperm(Bn, generator=g).to(torch.int32).to(dev())]
    res = []
    for od in orders:
        out = torch.zeros(1, rows, inner, dtype=T, device=dev())
        st = torch.zeros(Bn, H, Lmax, 4, device=dev())
        ops.attn_fwd(*args, out.unflatten(2, (H, dk)), st, order=od, **kw)
        dqkv = torch.zeros_like(qkv)
        drel = torch.zeros_like(rel)
        ops.attn_bwd(*args, out.unflatten(2, (H, dk)), heads(dout, 0), st, *(heads(dqkv, c) for c in (0, inner, 2 * inner)),
                     drel=drel, order=od, **{k: v for k, v in kw.items()})
        res.append((out, st.clone(), dqkv, drel))
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[2], res[0][2])
        valid = torch.zeros(Bn, Lmax, dtype=torch.bool, device=dev())
        for b, n in enumerate(lens):
            valid[b, :n] = True
        assert torch.equal(r[1][valid[:, None].expand(-1, H, -1)], res[0][1][valid[:, None].expand(-1, H, -1)])
        close(r[3], res[0][3], torch.float32, "bias gradient under a processing order", k=5)
    with pytest.raises(Exception):
        ops.attn_fwd(*args, torch.zeros(1, rows, inner, dtype=T, device=dev()).unflatten(2, (H, dk)), torch.zeros(Bn, H, Lmax, 4, device=dev()),
                     order=orders[1][:-1].contiguous(), **kw)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("kind", ["self", "self_drop", "cross"])
def test_attention_ragged_equals_padded(ops, dt, kind):
    """Ragged sequences (q_off / k_off over packed rows) must reproduce the PADDED call with a key mask on the rows that
    exist: outputs, softmax statistics, dq / dk / dv and the bias gradient — including an empty sequence."""
    T = DT[dt]
    H, dk = 4, 64
    inner = H * dk
    if kind == "cross":
        Bn, Lq, Lmax = 3, 6, 150
        lens = [150, 0 + 37, 96]
    else:
        Bn, Lmax = 5, 70
        Lq = Lmax
        lens = [70, 33, 0, 48, 1]
    drop = (0.1, 5, 9) if kind == "self_drop" else None
    off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    rows = int(off[-1])
    km = (torch.arange(Lmax, device=dev())[None] < torch.tensor(lens, device=dev())[:, None]).to(torch.uint8)
    rel = rnd(H, 2 * Lmax - 1, seed=44) if kind != "cross" else None
    bias = dict(rel_bias=rel, rel_off=Lmax - 1) if rel is not None else {}

    def pack(tp):       # [Bn, Lmax, C] padded → [1, rows, C] packed (existing rows only)
        return torch.cat([tp[b, :lens[b]] for b in range(Bn)], 0)[None].contiguous()

    if kind == "cross":
        q_p = rnd(Bn, Lq, inner, dtype=T, seed=45, scale=0.5)
        kv_p = rnd(Bn, Lmax, 2 * inner, dtype=T, seed=46, scale=0.5)
        kv_r = pack(kv_p)
        heads = lambda t, c0: t[:, :, c0:c0 + inner].unflatten(2, (H, dk))      # noqa: E731
        args_p = (heads(q_p, 0), heads(kv_p, 0), heads(kv_p, inner))
        args_r = (heads(q_p, 0), heads(kv_r, 0), heads(kv_r, inner))
        rag = dict(k_off=off, max_k=Lmax)
    else:
        qkv_p = rnd(Bn, Lmax, 3 * inner, dtype=T, seed=47, scale=0.5)
        qkv_r = pack(qkv_p)
        heads = lambda t, c0: t[:, :, c0:c0 + inner].unflatten(2, (H, dk))      # noqa: E731
        args_p = tuple(heads(qkv_p, c) for c in (0, inner, 2 * inner))
        args_r = tuple(heads(qkv_r, c) for c in (0, inner, 2 * inner))
        rag = dict(q_off=off, k_off=off, max_q=Lmax, max_k=Lmax)
    nq = Lq
    out_p = torch.zeros(Bn, nq, inner, dtype=T, device=dev())
    st_p = torch.zeros(Bn, H, nq, 4, device=dev())
    ops.attn_fwd(*args_p, out_p.unflatten(2, (H, dk)), st_p, key_mask=km, drop=drop, **bias)
    if kind == "cross":
        out_r, st_r = torch.zeros_like(out_p), torch.zeros_like(st_p)
        ops.attn_fwd(*args_r, out_r.unflatten(2, (H, dk)), st_r, drop=drop, **bias, **rag)
        assert torch.equal(out_r, out_p) and torch.equal(st_r[..., :2], st_p[..., :2])
    else:
        out_r = torch.zeros(1, rows, inner, dtype=T, device=dev())
        st_r = torch.zeros_like(st_p)
        ops.attn_fwd(*args_r, out_r.unflatten(2, (H, dk)), st_r, drop=drop, **bias, **rag)
        # bf16: the ragged launch runs the fast path (csrc/attn_enc.hip), the padded one (it carries a key mask) the generic kernels —
        # the same math in a different instruction order: equal to bf16 rounding, not bit for bit
        same = torch.equal if T == torch.float32 else (lambda x, y: (close(x, y, T, "ragged vs padded", k=0.5) or True))
        assert same(out_r, pack(out_p))
        for b in range(Bn):
            assert same(st_r[b, :, :lens[b], :2], st_p[b, :, :lens[b], :2]) if lens[b] else True
    # backward
    dout_p = rnd(Bn, nq, inner, dtype=T, seed=48)
    if kind != "cross":
        dout_p = dout_p * km[:, :, None].to(T)            # rows that do not exist carry no gradient
    drel_p = torch.zeros_like(rel) if rel is not None else None
    drel_r = torch.zeros_like(rel) if rel is not None else None
    if kind == "cross":
        dq_p, dkv_p = torch.zeros_like(q_p), torch.zeros_like(kv_p)
        ops.attn_bwd(*args_p, out_p.unflatten(2, (H, dk)), dout_p.unflatten(2, (H, dk)), st_p, heads(dq_p, 0), heads(dkv_p, 0),
                     heads(dkv_p, inner), key_mask=km, drop=drop)
        dq_r, dkv_r = torch.zeros_like(q_p), torch.zeros_like(kv_r)
        ops.attn_bwd(*args_r, out_r.unflatten(2, (H, dk)), dout_p.unflatten(2, (H, dk)), st_r, heads(dq_r, 0), heads(dkv_r, 0),
                     heads(dkv_r, inner), drop=drop, **rag)
        assert torch.equal(dq_r, dq_p) and torch.equal(dkv_r, pack(dkv_p))
    else:
        dqkv_p, dqkv_r = torch.zeros_like(qkv_p), torch.zeros_like(qkv_r)
        ops.attn_bwd(*args_p, out_p.unflatten(2, (H, dk)), dout_p.unflatten(2, (H, dk)), st_p, heads(dqkv_p, 0),
                     heads(dqkv_p, inner), heads(dqkv_p, 2 * inner), key_mask=km, drop=drop, drel=drel_p, **bias)
        ops.attn_bwd(*args_r, out_r.unflatten(2, (H, dk)), pack(dout_p).unflatten(2, (H, dk)), st_r, heads(dqkv_r, 0),
                     heads(dqkv_r, inner), heads(dqkv_r, 2 * inner), drop=drop, drel=drel_r, **bias, **rag)
        assert same(dqkv_r, pack(dqkv_p))
        close(drel_r, drel_p, T, "ragged drel", k=5)      # atomics: order differs


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("M,V", [(12, 64), (128, 32128)])
def test_cross_entropy(ops, ref, dt, M, V):
    T = DT[dt]
    logits = rnd(M, V, seed=26, scale=2.0)
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(4))
    labels[::3] = -100
    labels = labels.to(dev())
    lo, lor = torch.zeros(2, device=dev()), torch.zeros(2, device=dev())
    dl, dlr = torch.empty(M, V, dtype=T, device=dev()), torch.empty(M, V, device=dev())
    up = torch.tensor([0.5], device=dev())
    ops.ce_fwd_bwd(logits, labels, lo, dl, up)
    ref.ce_fwd_bwd(logits, labels, lor, dlr, up)
    assert abs(lo[0].item() - lor[0].item()) < 1e-4 * max(1.0, abs(lor[0].item())), (lo, lor)
    assert lo[1].item() == lor[1].item()
    close(dl, dlr, T, f"ce dlogits {dt}", tight=True)


def test_cross_entropy_non_finite_rows(ops):
    """a NaN / Inf row loss must show in the reported mean (the fixed-point sum alone would hide it: ADVICE round 2)"""
    M, V = 8, 512
    for poison, check in ((float("nan"), torch.isnan), (float("-inf"), lambda t: torch.isinf(t) & (t > 0))):
        logits = rnd(M, V, seed=33, scale=2.0)
        labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(5)).to(dev())
        logits[3, int(labels[3])] = poison          # NaN logit: NaN loss; -inf at the label: lse finite, loss +inf
        lo = torch.zeros(2, device=dev())
        ops.ce_fwd_bwd(logits, labels, lo, None)
        assert bool(check(lo[0])), (poison, lo)
    logits = rnd(M, V, seed=34, scale=2.0)          # finite rows still give the finite mean
    labels = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(6)).to(dev())
    lo = torch.zeros(2, device=dev())
    ops.ce_fwd_bwd(logits, labels, lo, None)
    want = torch.nn.functional.cross_entropy(logits.float().cpu(), labels.cpu())
    assert abs(lo[0].item() - want.item()) < 1e-4 * max(1.0, abs(want.item()))


def test_optimizer(ops, ref):
    n = 1000 * 4
    p, g = rnd(n, seed=27), rnd(n, seed=28, scale=3.0)
    m, v = rnd(n, seed=29, scale=0.1), rnd(n, seed=30).abs() * 0.01
    for T in (torch.float32, torch.bfloat16):
        for use_norm in (True, False):
            a = [t.clone() for t in (p, g, m, v)]
            b = [t.clone() for t in (p, g, m, v)]
            sh, shr = torch.empty(n, dtype=T, device=dev()), torch.empty(n, device=dev())
            ns, nsr = torch.zeros(1, device=dev()), torch.zeros(1, device=dev())
            ops.sumsq(a[1], ns)
            ref.sumsq(b[1], nsr)
            close(ns, nsr, torch.float32, "sumsq")
            kw = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=1e-2, max_norm=1.0, grad_scale=0.5)
            ops.adamw_step(a[0], a[1], a[2], a[3], sh, gnorm_sq=ns if use_norm else None, **kw)
            ref.adamw_step(b[0], b[1], b[2], b[3], shr, gnorm_sq=nsr if use_norm else None, **kw)
            for x, y, nm in zip(a, b, "pgmv"):
                close(x, y, torch.float32, f"adamw {nm}")
            close(sh, shr, T, "adamw shadow", tight=True)
    src = rnd(70, 200, seed=31)
    for T in (torch.float32, torch.bfloat16):
        dst = torch.empty(200, 70, dtype=T, device=dev())
        ops.transpose_cast(src, dst)
        assert torch.equal(dst, src.t().to(T))
        d2 = torch.empty(70 * 200, dtype=T, device=dev())
        ops.cast(src.view(-1), d2)
        assert torch.equal(d2, src.view(-1).to(T))
    # batched form: a table of matrices in one launch — ragged (non-multiple-of-64 / of-4) and aligned shapes
    shapes = [(70, 200), (64, 64), (3, 5), (128, 768), (257, 36), (1, 64)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += (r * c + 3) // 4 * 4
    flat = rnd(total, seed=32)
    desc, prefix, tiles = [], [], 0
    for (r, c), o in zip(shapes, offs):
        desc += [o, o, r, c]
        prefix.append(tiles)
        tiles += -(-r // 64) * -(-c // 64)
    for T in (torch.float32, torch.bfloat16):
        out = torch.full((total,), 7.0, dtype=T, device=dev())
        ops.transpose_cast_batched(flat, out, torch.tensor(desc, device=dev()), torch.tensor(prefix, dtype=torch.int32, device=dev()),
                                   len(shapes), tiles)
        for (r, c), o in zip(shapes, offs):
            assert torch.equal(out[o:o + r * c].view(c, r), flat[o:o + r * c].view(r, c).t().to(T)), (r, c, T)
    # from the bf16 shadow (what the training step does after the optimizer wrote it): the same copies from half the bytes
    out2 = torch.full((total,), 7.0, dtype=torch.bfloat16, device=dev())
    ops.transpose_cast_batched(flat.bfloat16(), out2, torch.tensor(desc, device=dev()), torch.tensor(prefix, dtype=torch.int32, device=dev()),
                               len(shapes), tiles)
    assert torch.equal(out2, out)


def test_pack_ids(ops, ref):
    """lako_pack_ids: the valid positions of a padded [BN, L] id batch in packed (passage, position) order — empty and full passages"""
    g = torch.Generator().manual_seed(3)
    BN, L = 37, 50
    lens = torch.randint(0, L + 1, (BN,), generator=g)
    lens[0], lens[5] = L, 0
    off = torch.zeros(BN + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(lens, 0)
    ids = torch.randint(0, 32000, (BN * L,), generator=g)
    out = torch.full((int(off[-1]),), -7, dtype=torch.int64, device=dev())
    outr = torch.full_like(out, -7)
    ops.pack_ids(ids.to(dev()), off.to(dev()), out, L)
    ref.pack_ids(ids.to(dev()), off.to(dev()), outr, L)
    assert torch.equal(out, outr) and int((out == -7).sum()) == 0


def test_int_helpers(ops, ref):
    labels = torch.tensor([[5, 9, 1, -100], [7, 1, -100, -100]], device=dev())
    dec = torch.full_like(labels, 99)
    ops.shift_right(labels, dec)
    assert dec.tolist() == [[0, 5, 9, 1], [0, 7, 1, 0]]
    B, V = 5, 32128
    logits = rnd(B, V, seed=32)
    logits[2, 100] = logits[2, 7000] = 50.0           # tie → lowest index (torch.argmax semantics)
    logits[3, 1] = 60.0                                # EOS
    seq = torch.zeros(B, 6, dtype=torch.long, device=dev())
    nxt = torch.zeros(B, dtype=torch.long, device=dev())
    done = torch.zeros(B, dtype=torch.uint8, device=dev())
    done[4] = 1
    nd = torch.zeros(1, dtype=torch.int32, device=dev())
    ops.greedy_step(logits, seq, 2, nxt, done, nd)
    exp = logits.argmax(-1)
    exp[4] = 0
    assert nxt.tolist() == exp.tolist() and nxt[2].item() == 100 and nxt[3].item() == 1
    assert seq[:, 2].tolist() == exp.tolist() and done.tolist() == [0, 0, 0, 1, 1] and nd.item() == 2


@pytest.mark.gpu
@pytest.mark.parametrize("style", ["mean", "max", "21mean"])
@pytest.mark.parametrize("half", [False, True])
def test_fact_scores_segmented_reduce(ops, ref, style, half):
    """lako_fact_scores (per-fact aggregation of captured cross-attention scores, src/model.py:100-115,143-204) against the
    reference's own loops (test double): terminated and unterminated last spans, a passage without any '.', padded tails,
    more spans than n_context, ties inside a span (21mean picks by rank)."""
    B, H, nl, N, L, n_ctx = 6, 4, 6, 2, 48, 5
    g = torch.Generator().manual_seed(11)
    scores = torch.randn(B, H, nl, N * L, generator=g) * 3
    scores[2, :, :, L + 4:L + 9] = 0.25                                   # ties
    ids = torch.randint(11, 60, (B, N, L), generator=g)
    mask = torch.ones(B, N, L, dtype=torch.bool)
    for b, (dots, valid) in enumerate([((7, 15, 22, 40), 41), ((9, 20), 30), ((), 48), ((5, 6, 7, 12, 19, 25, 33), 48),
                                       ((10, 47), 48), ((3,), 4)]):
        ids[b, 1, list(dots)] = 5
        ids[b, 1, valid:] = 0
        mask[b, 1, valid:] = False
    layer0 = nl - nl // 2 if half else 0
    want = torch.zeros(B, n_ctx, dtype=torch.float64)
    ref.fact_scores(scores, mask, ids, want, layer0=layer0, layers_used=nl - layer0, passage=1, style=style)
    got = torch.zeros(B, n_ctx, dtype=torch.float64, device=dev())
    ops.fact_scores(scores.to(dev()), mask.to(dev()).to(torch.uint8), ids.to(dev()), got, layer0=layer0, layers_used=nl - layer0,
                    passage=1, style=style)
    # the per-position head/layer sums are rounded to fp32 in a different order (torch.sum vs an fp64 accumulation)
    torch.testing.assert_close(got.cpu(), want, atol=2e-6, rtol=2e-6)
    assert (got.cpu()[2] == -5.0 / ((nl - layer0) * H))[1:].all()          # no '.', not padded → ONE span, then the −5 filler
    assert (got.cpu()[5] == -5.0 / ((nl - layer0) * H))[1:].all()
    # N > 2: the reference takes the SCORES of passage ceil(N/2) and the TOKEN IDS of passage 1 (src/model.py:164-174)
    N4 = 4
    scores4 = torch.randn(B, H, nl, N4 * L, generator=g) * 3
    ids4 = torch.randint(11, 60, (B, N4, L), generator=g)
    ids4[:, 1] = ids[:, 1]
    mask4 = torch.ones(B, N4, L, dtype=torch.bool)
    mask4[:, 2, 40:] = False
    want4 = torch.zeros(B, n_ctx, dtype=torch.float64)
    ref.fact_scores(scores4, mask4, ids4, want4, layer0=layer0, layers_used=nl - layer0, passage=2, ids_passage=1, style=style)
    got4 = torch.zeros(B, n_ctx, dtype=torch.float64, device=dev())
    ops.fact_scores(scores4.to(dev()), mask4.to(dev()).to(torch.uint8), ids4.to(dev()), got4, layer0=layer0, layers_used=nl - layer0,
                    passage=2, ids_passage=1, style=style)
    torch.testing.assert_close(got4.cpu(), want4, atol=2e-6, rtol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_attention_dropout_mask_matches_recipe(ops, dt):
    """The attention kernels (generic fp32 path and the bf16 fast path) draw exactly the keep mask of the block-hash recipe
    (csrc/attn_shared.h ↔ tests/ref_ops.attn_keep_mask): with q = 0 every probability is 1/Lk, and one-hot value rows make
    out[q][d] = keep[q][d]·scale/Lk — the mask itself, 64 keys at a time.  The backward must regenerate the same mask in
    both of its passes (dQ: 4 keys of one query per lane; dK/dV: 4 queries of one key per lane)."""
    from tests.ref_ops import attn_keep_mask, drop_key
    T = DT[dt]
    Bn, H, L, dk = 3, 2, 64, 64
    drop = (0.25, 77, 5)
    q = torch.zeros(Bn, L, H, dk, dtype=T, device=dev())
    k = rnd(Bn, L, H, dk, dtype=T, seed=3)
    v = torch.eye(L, dk, device=dev())[None, :, None, :].expand(Bn, L, H, dk).to(T).contiguous()
    out = torch.zeros(Bn, L, H, dk, dtype=T, device=dev())
    st = torch.zeros(Bn, H, L, 4, device=dev())
    ops.attn_fwd(q, k, v, out, st, drop=drop)
    keep = attn_keep_mask(Bn * H, L, L, drop_key(drop[1], drop[2]), drop[0]).view(Bn, H, L, L)
    got = (out.float().cpu() > 0).permute(0, 2, 1, 3)          # [Bn, H, q, key]
    assert torch.equal(got, keep)
    assert abs(float(keep.double().mean()) - 0.75) < 0.01
    # backward: dV[key][d] = Σ_q keep[q,key]·scale/L·dO[q][d]; with dO = one-hot(q → d): dV[key][d] = keep[d, key]·scale/L
    dout = torch.eye(L, dk, device=dev())[None, :, None, :].expand(Bn, L, H, dk).to(T).contiguous()
    dq, dk_, dv = (torch.zeros_like(t) for t in (q, k, v))
    ops.attn_bwd(q, k, v, out, dout, st, dq, dk_, dv, drop=drop)
    got_v = (dv.float().cpu() > 0).permute(0, 2, 3, 1)         # [Bn, H, d = query, key]
    assert torch.equal(got_v, keep)


def test_attention_dropout_recipe_statistics():
    """Keep rate and neighbour correlations of the block-hash recipe at p = 0.1 (CPU, the integer recipe of tests/ref_ops.py that
    the kernels are checked against bit for bit on the GPU)."""
    from tests.ref_ops import attn_keep_mask
    for key in (0x12345678, 0x9E3779B9):
        k = attn_keep_mask(96, 200, 200, key, 0.1).double()
        assert abs(float(k.mean()) - 0.9) < 6e-4
        c = lambda a, b: float(np.corrcoef(a.reshape(-1).numpy(), b.reshape(-1).numpy())[0, 1])   # noqa: E731
        for a_, b_ in ((k[:, :, :-1], k[:, :, 1:]), (k[:, :, :-2], k[:, :, 2:]), (k[:, :-1], k[:, 1:]), (k[:, :-2], k[:, 2:]),
                       (k[:, :-4], k[:, 4:]), (k[:, :, :-4], k[:, :, 4:]), (k[:-1], k[1:]), (k[:, :-1, :-1], k[:, 1:, 1:])):
            assert abs(c(a_, b_)) < 2.5e-3
        assert abs(float(k.mean(2).std()) - (0.09 / 200) ** 0.5) < 1.5e-3          # per-row keep fractions: binomial spread


# ------------------------------------------------------------------------------------------------
# MX block-scaled fp8 GEMM (BASELINE config 5)
@pytest.mark.gpu
@pytest.mark.parametrize("rows,K", [(300, 768), (64, 3072), (1000, 128), (33, 1024)])
def test_mx_quantize(ops, rows, K):
    """lako_mx_quantize against the recipe in tests/ref_ops.py: e4m3 bytes and E8M0 block scales bit for bit (incl. an all-zero
    block, a block with one huge value, tiny values) and the [rows, 4, KSP] scale layout."""
    from tests.ref_ops import mx_quantize_ref, mx_scales_layout
    x = rnd(rows, K, seed=5, scale=2.0)
    x[3, 32:64] = 0.0
    x[5, 7] = 3.0e4
    x[7, 96:128] *= 1e-6
    xb = x.to(torch.bfloat16)
    q = torch.zeros(rows, K, dtype=torch.uint8, device=dev())
    sc = torch.zeros(rows, ops.mx_scale_cols(K), dtype=torch.uint8, device=dev())
    ops.mx_quantize(xb, q, sc)
    qr, ex = mx_quantize_ref(xb.float().cpu())
    assert torch.equal(q.cpu().view(torch.float8_e4m3fn).float(), qr)
    assert torch.equal(sc.cpu(), mx_scales_layout(ex, K))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,d", [(300, 1024), (1000, 768), (65, 128), (4099, 512)])
def test_rmsnorm_fwd_mx_equals_norm_then_quantise(ops, rows, d):
    """lako_rmsnorm_fwd_mx (round 6): the RMSNorm forward with the MX quantiser of its output fused in — y, rstd, the e4m3 bytes and the scale
    bytes are BIT-identical to lako_rmsnorm_fwd followed by lako_mx_quantize (the quantiser reads the rounded bf16 row either way); a row
    holding a NaN and a row of zeros included."""
    T = torch.bfloat16
    x = rnd(rows, d, dtype=T, seed=51, scale=3.0)
    x[3, 7] = float("nan")
    x[5] = 0
    w = rnd(d, seed=52) + 1.0
    y0, y1 = (torch.zeros(rows, d, dtype=T, device=dev()) for _ in range(2))
    r0, r1 = (torch.zeros(rows, device=dev()) for _ in range(2))
    q0, q1 = (torch.zeros(rows, d, dtype=torch.uint8, device=dev()) for _ in range(2))
    s0, s1 = (torch.zeros(rows, ops.mx_scale_cols(d), dtype=torch.uint8, device=dev()) for _ in range(2))
    ops.rmsnorm_fwd(x, w, y0, r0, 1e-6)
    ops.mx_quantize(y0, q0, s0)
    ops.rmsnorm_fwd_mx(x, w, y1, r1, 1e-6, q1, s1)
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16)) and torch.equal(r0.view(torch.int32), r1.view(torch.int32))
    assert torch.equal(q0, q1) and torch.equal(s0, s1)
    assert int(s1[3].max()) == 0xFF and int((q1[5] & 0x7F).max()) == 0      # (a NaN row: every block's scale is the E8M0 NaN; zeros: ±0)


@pytest.mark.gpu
def test_mx_quantize_propagates_nan(ops):
    """a NaN activation must stay visible in the fp8 forward: its element becomes the e4m3 NaN (0x7F), its
    block's scale the E8M0 NaN (0xFF), the other blocks are untouched, and the product rows that meet the block are NaN"""
    from tests.ref_ops import mx_quantize_ref, mx_scales_layout
    rows, K = 64, 256
    x = rnd(rows, K, seed=8, scale=2.0).to(torch.bfloat16)
    x[5, 40] = float("nan")
    q = torch.zeros(rows, K, dtype=torch.uint8, device=dev())
    sc = torch.zeros(rows, ops.mx_scale_cols(K), dtype=torch.uint8, device=dev())
    ops.mx_quantize(x, q, sc)
    clean = x.clone()
    clean[5, 32:64] = 0.0
    qr, ex = mx_quantize_ref(clean.float().cpu())
    want_sc = mx_scales_layout(ex, K).view(rows, 4, -1)
    want_sc[5, 1, 0] = 0xFF                                  # block 1 of row 5
    assert torch.equal(sc.cpu().view(rows, 4, -1), want_sc)
    assert int(q[5, 40]) == 0x7F
    keep = torch.ones(rows, K, dtype=torch.bool)
    keep[5, 32:64] = False
    assert torch.equal(q.cpu().view(torch.float8_e4m3fn).float()[keep], qr[keep])
    B = rnd(256, K, seed=9).to(torch.bfloat16)
    Bq = torch.zeros(256, K, dtype=torch.uint8, device=dev())
    Bs = torch.zeros(256, ops.mx_scale_cols(K), dtype=torch.uint8, device=dev())
    ops.mx_quantize(B, Bq, Bs)
    Cm = torch.zeros(rows, 256, dtype=torch.bfloat16, device=dev())
    ops.gemm_nt_mx(q, sc, Bq, Bs, Cm)
    assert bool(torch.isnan(Cm[5].float()).all()) and not bool(torch.isnan(Cm[:5].float()).any()) and not bool(torch.isnan(Cm[6:].float()).any())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 264, 768), (1024, 768, 3072), (3000, 2304, 768), (70000, 768, 768)])
def test_gemm_nt_mx(ops, M, N, K):
    """lako_gemm_nt_mx: quantise both operands with lako_mx_quantize, multiply on the block-scaled matrix cores, compare with the
    fp32 product of the DEQUANTISED operands (the kernel's only rounding is fp32 accumulation + the bf16 store) — edge tiles,
    several K-step groups (scale dword reloads), more tiles than workgroups (persistent loop)."""
    from tests.ref_ops import mx_dequant_ref, mx_quantize_ref
    A = rnd(M, K, seed=31, scale=1.5).to(torch.bfloat16)
    B = rnd(N, K, seed=32, scale=0.7).to(torch.bfloat16)
    Aq, Bq = (torch.zeros(t.shape, dtype=torch.uint8, device=dev()) for t in (A, B))
    As, Bs = (torch.zeros(t.shape[0], ops.mx_scale_cols(K), dtype=torch.uint8, device=dev()) for t in (A, B))
    ops.mx_quantize(A, Aq, As)
    ops.mx_quantize(B, Bq, Bs)
    Cm = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
    ops.gemm_nt_mx(Aq, As, Bq, Bs, Cm, alpha=0.5)
    Ad = mx_dequant_ref(*mx_quantize_ref(A.float().cpu())).to(dev())
    Bd = mx_dequant_ref(*mx_quantize_ref(B.float().cpu())).to(dev())
    want = 0.5 * (Ad @ Bd.t())
    close(Cm, want, torch.bfloat16, f"gemm_nt_mx {M}x{N}x{K}", k=0.5)
    # and it is a usable approximation of the bf16 product: block-scaled e4m3 carries ≈ 2^-4 relative error per element
    exact = 0.5 * (A.float() @ B.float().t())
    rel = float((Cm.float() - exact).norm() / exact.norm())
    assert rel < 0.06, rel


@pytest.mark.gpu
def test_gemm_nt_mx_epilogues(ops, ref):
    from tests.ref_ops import mx_dequant_ref, mx_quantize_ref
    M, N, K = 520, 768, 768
    A = rnd(M, K, seed=41).to(torch.bfloat16)
    B = rnd(N, K, seed=42, scale=0.5).to(torch.bfloat16)
    res = rnd(M, N, seed=43).to(torch.bfloat16)
    aux = rnd(M, N, seed=44).to(torch.bfloat16)
    Aq, Bq = (torch.zeros(t.shape, dtype=torch.uint8, device=dev()) for t in (A, B))
    As, Bs = (torch.zeros(t.shape[0], ops.mx_scale_cols(K), dtype=torch.uint8, device=dev()) for t in (A, B))
    ops.mx_quantize(A, Aq, As)
    ops.mx_quantize(B, Bq, Bs)
    Ad = mx_dequant_ref(*mx_quantize_ref(A.float().cpu())).to(dev()).to(torch.bfloat16)
    Bd = mx_dequant_ref(*mx_quantize_ref(B.float().cpu())).to(dev()).to(torch.bfloat16)
    for kw in (dict(relu=True, drop=(0.1, 3, 4)), dict(resid=res, drop=(0.1, 5, 6)), dict(aux=aux, aux_scale=1.0 / 0.9), dict()):
        got = torch.zeros(M, N, dtype=torch.bfloat16, device=dev())
        want = torch.zeros(M, N, device=dev())
        ops.gemm_nt_mx(Aq, As, Bq, Bs, got, **kw)
        ref.gemm_nt(Ad, Bd, want, **kw)                    # same epilogue recipe on the dequantised (bf16-exact) operands
        close(got, want, torch.bfloat16, f"gemm_nt_mx epilogue {sorted(kw)}", k=0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(70000, 768, 1024), (40000, 1032, 512), (50001, 520, 2048), (16500, 3072, 1024)])
def test_gemm_nt_mx_four_wave(ops, ref, M, N, K):
    """(round 6, csrc/gemm_nt4_mx.h) The MX product on the four-wave tile — two K-slices of LDS-DMA in flight behind a counted wait, scale bytes
    picked by op_sel from a dword per group of four K-steps, nt4_epilogue — forced to 256-row and 192-row tiles and as the default plan, against
    the eight-wave MX kernel (`gemm_nt_four` 0): BIT-identical outputs (the same MFMAs in the same K order per element, the same epilogue
    arithmetic) for every epilogue, ragged last tiles in M and N, several tiles per workgroup, one and several scale groups per tile; and
    against the dequantised fp32 product."""
    from tests.ref_ops import mx_dequant_ref, mx_quantize_ref
    T = torch.bfloat16
    A, B = rnd(M, K, seed=91).to(T), rnd(N, K, seed=92, scale=0.6).to(T)
    R, X = rnd(M, N, dtype=T, seed=93), rnd(M, N, dtype=T, seed=94)
    Aq, Bq = (torch.zeros(t.shape, dtype=torch.uint8, device=dev()) for t in (A, B))
    As, Bs = (torch.zeros(t.shape[0], ops.mx_scale_cols(K), dtype=torch.uint8, device=dev()) for t in (A, B))
    ops.mx_quantize(A, Aq, As)
    ops.mx_quantize(B, Bq, Bs)
    try:
        for kw in (dict(), dict(relu=True, drop=(0.1, 5, 6)), dict(alpha=0.5), dict(resid=R, drop=(0.1, 7, 8)), dict(resid=R), dict(aux=X, aux_scale=1.1)):
            got = {}
            for name, four, variant in (("eight-wave", 0, -1), ("default plan", 1, -1), ("256-row", 1, 9), ("192-row", 1, 3)):
                ops.set_tuning("gemm_nt_four", four)
                ops.set_tuning("gemm_nt_variant", variant)
                C = torch.full((M, N), float("nan"), dtype=T, device=dev())
                ops.gemm_nt_mx(Aq, As, Bq, Bs, C, **kw)
                torch.cuda.synchronize()
                got[name] = C
            for name in got:
                assert torch.equal(got[name].view(torch.int16), got["eight-wave"].view(torch.int16)), f"{name} vs the eight-wave MX kernel, {list(kw)} {M}x{N}x{K}"
        if M <= 20000:
            Ad = mx_dequant_ref(*mx_quantize_ref(A.float().cpu())).to(dev())
            Bd = mx_dequant_ref(*mx_quantize_ref(B.float().cpu())).to(dev())
            close(got["default plan"], (Ad @ Bd.t()) * 1.1 * (X.float() > 0), T, f"gemm_nt_mx four-wave {M}x{N}x{K}", k=0.5)
    finally:
        ops.set_tuning("gemm_nt_four", 1)
        ops.set_tuning("gemm_nt_variant", -1)

