"""GPU unit tests of the primitive HIP kernels (called through the C ABI) against plain PyTorch fp32 on the CPU."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from iisan_amd import _lib  # noqa: E402

T16 = {0: torch.float16, 1: torch.bfloat16}
TOL = {0: 2e-3, 1: 1.6e-2}       # relative to the output scale: one 16-bit rounding of the result


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _pad_rows(t, mult=256):
    M = t.shape[0]
    Mp = (M + mult - 1) // mult * mult
    out = torch.zeros((Mp,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    out[:M] = t
    return out


@pytest.mark.parametrize("variant", [1, 3, 4])
@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(300, 256, 192), (128, 128, 64), (1000, 768, 3072), (517, 2304, 768), (70000, 768, 768)])
def test_gemm16_vs_torch(lib, variant, dt, mode, shape):
    """variant 1 = 128x128 v1 kernel, 2 = persistent lock-step 256x256 kernel, 3 = persistent staggered 256x256 kernel
    (16-bit epilogues only); 2 and 3 fall back when the shape does not qualify (N % 256 != 0, K < 128)."""
    M, N, K = shape
    _lib.dev_set("gemm16_variant", variant)
    g = torch.Generator().manual_seed(M * 7 + N + K + mode)
    # asymmetric, non-identity operands (catches transposed / permuted fragment layouts)
    A = (torch.randn(M, K, generator=g) * 0.5).to(T16[dt])
    W = (torch.randn(N, K, generator=g) * 0.05 + torch.linspace(-0.02, 0.03, N)[:, None]).to(T16[dt])
    bias = torch.randn(N, generator=g) * 0.3
    resid = torch.randn(M, N, generator=g)
    ref = A.double() @ W.double().t() + bias.double()
    if mode == 1:
        ref = torch.nn.functional.gelu(ref)
    if mode == 2:
        ref = ref + resid.double()
    Ad, Wd, bd = _pad_rows(A.cuda()), W.cuda(), bias.cuda()
    if mode == 2:
        out = resid.cuda().clone()            # in-place residual, as the encoders use it
        rp = out
    else:
        out = torch.empty(M, N, dtype=T16[dt], device="cuda")
        rp = None
    _lib.check(lib.iisan_gemm16(dt, mode, Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(),
                                rp.data_ptr() if rp is not None else None, M, N, K, _stream()), "gemm16")
    torch.cuda.synchronize()
    _lib.dev_set("gemm16_variant", 0)
    got = out.cpu().double()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    tol = (2e-5 if mode == 2 else TOL[dt]) * scale
    assert err <= tol, f"gemm16 variant={variant} dt={dt} mode={mode} {shape}: max err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("shape", [(70000, 2304, 768, 0), (42240, 3072, 768, 1), (66000, 768, 3072, 0), (1300, 768, 256, 0),
                                   (40000, 768, 128, 0), (40000, 1536, 192, 1)])
def test_gemm16_h256_race_screen_against_the_s256_kernel(lib, shape):
    """Race screen of the half-slot tile boundary (`csrc/gemm16_h256.hip`): its LDS-DMA pieces are issued from other slots than
    in `gemm16_s256.hip`, and a piece read before it has landed gives rare wrong tiles that come and go with memory load
    (cdna_hip_programming.md: place reads by the vmcnt / barrier count, never by clean runs).  Both kernels are deterministic
    and accumulate in the same order, so every run of either must give the same bits: 20 runs on fresh operands each, many
    tiles per workgroup (70,000 rows x 9 column tiles = 2,466 tiles on 256 CUs), a ragged last row tile, K = 768 and 3072,
    plain and GELU epilogues — and the SHORTEST tiles the kernel accepts: K = 128 (a tile is a first and a last K-step
    with no middle step, every plan crosses the tile boundary) and K = 192 (one middle step); the panel tile walk of round 4 is
    what the auto policy picks for the 2304- and 3072-column shapes here when M >= 32768."""
    M, N, K, mode = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    out = {v: torch.empty(M, N, dtype=torch.float16, device="cuda") for v in (3, 4)}
    for it in range(20):
        A = _pad_rows((torch.randn(M, K, generator=g, device="cuda") * 0.5).half())
        W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
        bias = torch.randn(N, generator=g, device="cuda") * 0.3
        for v in (3, 4):
            _lib.dev_set("gemm16_variant", v)
            out[v].fill_(float("nan"))
            _lib.check(lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out[v].data_ptr(), None, M, N, K, _stream()), "gemm16")
        _lib.dev_set("gemm16_variant", 0)
        torch.cuda.synchronize()
        assert torch.isfinite(out[4]).all(), f"iteration {it}: rows not written"
        assert torch.equal(out[3], out[4]), f"iteration {it}: {(out[3].float() - out[4].float()).abs().max().item():.3e}"


@pytest.mark.parametrize("shape", [(2304, 5000, 2304, 256, 4), (6000, 6000, 1536, 192, 1), (2048, 2050, 3072, 128, 0)])
def test_gemm16_h256_tile_walks_are_bit_identical(lib, shape):
    """Round 4: `gemm16_h256_kernel` walks the tile space per XCD in panels (dev switch `gemm16_walk` (c, h): c column tiles wide, sub-slabs of
    h row tiles; the auto policy uses 3 x 16 for N >= 1536, M >= 32768).  The walk only changes WHICH workgroup computes a tile and when:
    every walk must give the bits of the row-major list — panels that do not divide the tile row (c = 2, 5 on 9 / 6 / 12 column tiles), a
    last sub-slab shorter than h, slabs of unequal height (row tiles not a multiple of 8), a ragged last row tile, fewer tiles than CUs
    in some slabs, K = 128 (every DMA plan crosses a tile boundary) — for the head-major QKV scatter, GELU and plain epilogues."""
    _, M, N, K, mode = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = _pad_rows((torch.randn(M, K, generator=g, device="cuda") * 0.5).half())
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    bias = torch.randn(N, generator=g, device="cuda") * 0.3
    ref = None
    try:
        for c, h in ((0, 0), (1, 1), (2, 3), (3, 16), (5, 7), (8, 2), (4, 0)):
            _lib.dev_set("gemm16_variant", 4)
            (_lib.dev_set("gemm16_walk_c", c), _lib.dev_set("gemm16_walk_h", h))
            if mode == 4:      # head-major QKV needs the executor's argument set: through the public entry the plain epilogue stands in
                out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
                _lib.check(lib.iisan_gemm16(0, 0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), None, M, N, K, _stream()), "gemm16")
            else:
                out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
                _lib.check(lib.iisan_gemm16(0, mode, A.data_ptr(), W.data_ptr(), bias.data_ptr(), out.data_ptr(), None, M, N, K, _stream()), "gemm16")
            torch.cuda.synchronize()
            assert torch.isfinite(out).all(), (c, h)
            if ref is None:
                ref = out
                want = A[:M].float() @ W.float().t() + bias
                if mode == 1:
                    want = torch.nn.functional.gelu(want)
                assert (out.float() - want).abs().max().item() <= 3e-3 * want.abs().max().item()
            else:
                assert torch.equal(out, ref), (c, h, (out.float() - ref.float()).abs().max().item())
    finally:
        _lib.dev_set("gemm16_variant", 0)
        (_lib.dev_set("gemm16_walk_c", -1), _lib.dev_set("gemm16_walk_h", 0))


def _ln_fold_case(M, N, K, row_mean, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, K, generator=g, device="cuda") * 0.8 + row_mean
    x[:, 7] += 20.0; x[:, 300] -= 9.0                        # the "massive activation" channels of a ViT stream
    x16 = _pad_rows(x.half())
    gamma = 1.0 + 0.3 * torch.randn(K, generator=g, device="cuda")
    beta = 0.2 * torch.randn(K, generator=g, device="cuda")
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).half()
    b = torch.randn(N, generator=g, device="cuda")
    xf = x16.float()
    mean = xf.mean(1, keepdim=True)
    rstd = torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-6)
    ln32 = (xf - mean) * rstd * gamma + beta
    return x16, gamma, beta, W, b, rstd.reshape(-1).contiguous(), ln32


@pytest.mark.parametrize("w32", [1, 0])
@pytest.mark.parametrize("row_mean", [0.3, 3.0])
def test_layernorm_fold_of_the_weights_is_centred_and_sum_preserving(lib, row_mean, w32):
    """Round 4 (second half): `fold_ln_weights_kernel` (rowops.hip) — Wf[n] = fp16(gamma * W[n] - mean_k(gamma * W[n])), bias' = b + W beta, from the
    caller's fp32 master (w32) or from the fp16 copy.  Sum-preserving rounding: every folded element within ONE ulp of its target with
    the rms of plain rounding (a handful per row are moved to their other neighbour), the ROW SUMS within one ulp of one weight of
    zero (plain rounding leaves ~20x that: the sum is what multiplies the mean of an activation row), and LN(x) W^T + b ==
    rstd * (x Wf^T) + bias' to fp16-operand accuracy in fp64 arithmetic — also for rows whose mean is several standard deviations."""
    N, K = 2304, 768
    x16, gamma, beta, W, b, rstd, ln32 = _ln_fold_case(512, N, K, row_mean, 11)
    W32 = W.float() * (1.0 + 2.0 ** -13) if w32 else None                 # a master that is NOT fp16-representable
    Wsrc = W32 if w32 else W
    Wf = torch.empty(N, K, dtype=torch.float16, device="cuda"); bf = torch.empty(N, device="cuda")
    _lib.check(lib.iisan_fold_ln_weights(Wsrc.data_ptr(), w32, b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), Wf.data_ptr(), bf.data_ptr(), N, _stream()), "fold")
    torch.cuda.synchronize()
    Wg = Wsrc.double() * gamma.double()
    Wc = Wg - Wg.mean(1, keepdim=True)
    ulp = 2.0 ** -10 * Wc.abs().clamp_min(2.0 ** -14)                     # fp16 spacing at the element's magnitude (upper bound)
    err = (Wf.double() - Wc).abs()
    assert (err[:, 1:] <= 1.001 * ulp[:, 1:]).all()
    # column 0 takes what the bisection leaves: under one ulp of the row's LARGEST weights, whatever its own size
    assert (err[:, 0] <= 1.5 * 2.0 ** -10 * Wc.abs().max(1).values).all()
    plain = Wc.half().double()
    assert err[:, 1:].pow(2).mean().sqrt().item() < 1.1 * (plain - Wc)[:, 1:].pow(2).mean().sqrt().item()
    sums = Wf.double().sum(1).abs()
    assert sums.max().item() <= 2.0 ** -10 * Wc.abs().max().item(), sums.max().item()
    assert sums.max().item() < 0.2 * plain.sum(1).abs().max().item(), (sums.max().item(), plain.sum(1).abs().max().item())
    assert (bf.double() - (b.double() + Wsrc.double() @ beta.double())).abs().max().item() < 1e-5
    ref = ((x16.double() - x16.double().mean(1, keepdim=True)) * rstd.double()[:, None] * gamma.double() + beta.double()) @ Wsrc.double().t() + b.double()
    alg = rstd.double()[:, None] * (x16.double() @ Wf.double().t()) + bf.double()
    e_alg = ((alg - ref).norm() / ref.norm()).item()
    assert e_alg < 2.2e-4, e_alg            # one rounding of the weights (measured ~1.6e-4: what rounding LayerNorm(x) to fp16 costs the other route)


@pytest.mark.parametrize("case", [("qkv", 4, 2304, 197 * 335), ("fc1", 1, 3072, 277376), ("fc1", 1, 3072, 1500)])
def test_gemm16_h256_layernorm_epilogue_every_element_every_run(lib, case):
    """Round 4 (second half): `gemm16_h256_kernel<.., LNA = true>` — LayerNorm applied by the epilogue of the product that consumes it: A = the
    un-normalised fp16 rows, W = the folded weights, one rstd per row (brought in per tile by LDS-DMA), out = rstd * acc + bias'.
    EVERY element is held to fp32 arithmetic (GELU) / to the same kernel on the materialised LayerNorm image (head-major QKV) over
    three launches that must agree bit for bit: the first version of this epilogue let the compiler broadcast the statistic by
    op_sel and lost a product in the last 16 lanes of a wave a few thousand times per 8.5e8 outputs, different ones every run."""
    name, mode, N, M = case
    K, S = 768, 197
    x16, gamma, beta, W, b, rstd, ln32 = _ln_fold_case(M, N, K, 0.3, N + M)
    Wf = torch.empty_like(W); bf = torch.empty(N, device="cuda")
    _lib.check(lib.iisan_fold_ln_weights(W.data_ptr(), 0, b.data_ptr(), gamma.data_ptr(), beta.data_ptr(), Wf.data_ptr(), bf.data_ptr(), N, _stream()), "fold")
    Mp = x16.shape[0]
    outs = []
    try:
        _lib.dev_set("gemm16_variant", 4)
        for rep in range(3):
            out = torch.zeros(Mp, N, dtype=torch.float16, device="cuda")
            _lib.check(lib.iisan_gemm16_lna(mode, x16.data_ptr(), Wf.data_ptr(), bf.data_ptr(), out.data_ptr(), rstd.data_ptr(), M, N, K, S, _stream()), "gemm16_lna")
            outs.append(out)
        img = torch.zeros(Mp, N, dtype=torch.float16, device="cuda")
        ln16 = ln32.half().contiguous()
        _lib.check(lib.iisan_gemm16_lna(mode, ln16.data_ptr(), W.data_ptr(), b.data_ptr(), img.data_ptr(), None, M, N, K, S, _stream()), "gemm16 (image)")
    finally:
        _lib.dev_set("gemm16_variant", 0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    worst = 0.0
    for r0 in range(0, M, 32768):
        r1 = min(M, r0 + 32768)
        if mode == 1:
            want = torch.nn.functional.gelu(ln32[r0:r1] @ W.float().t() + b)
        else:
            want = img[r0:r1].float()
        worst = max(worst, (outs[0][r0:r1].float() - want).abs().max().item())
    assert worst < 0.02, worst          # outputs are O(1..10); a lost product is O(0.1..1), 16-bit rounding of both sides 8e-3
    if mode == 1:
        assert not outs[0][M:].any()    # rows beyond M are not written


@pytest.mark.parametrize("case", [(768, 1408), (3072, 352), (768, 9), (128, 1408), (192, 40)])
def test_gemm16_h256_stream_epilogue_adds_into_the_stream_and_leaves_row_sums(lib, case):
    """Round 4 (second half): `EPI_STREAM16` — the residual add of a pre-LN tower in the epilogue of the O / FC2 products: x[m] = fp16(x[m] + A[m] W^T
    + b) in place in the fp16 stream, (sum, sum of squares) of every new row over each 64-column slice, the CLS rows (m = item * S)
    receive the delta alone; `stream_stats_finalize` turns the sums into rstd in a fixed order and folds the CLS deltas into their
    fp32 stream.  Every element against fp32 arithmetic, the partial sums against torch, three launches bit-identical; K = 128 / 192:
    tiles of two / three K-steps (no middle step between the two prefetches of old stream values), fewer tiles than CUs."""
    K, items = case
    S, N = 197, 768
    M = items * S
    g = torch.Generator(device="cuda").manual_seed(K + items)
    A = _pad_rows((torch.randn(M, K, generator=g, device="cuda") * 0.5).half())
    W = (torch.randn(N, K, generator=g, device="cuda") * (0.025 if K == 3072 else 0.05)).half()
    b = torch.randn(N, generator=g, device="cuda") * 0.3
    x0 = _pad_rows((torch.randn(M, N, generator=g, device="cuda") * 1.5 + 0.2).half())
    x0[:, 7] += 20.0
    xc0 = torch.randn(items, N, generator=g, device="cuda") * 1.5
    Mp = A.shape[0]
    runs = []
    try:
        _lib.dev_set("gemm16_variant", 4)
        for rep in range(3):
            x = x0.clone(); xc = xc0.clone()
            part = torch.zeros(N // 64, Mp, 2, device="cuda"); rstat = torch.zeros(Mp, device="cuda")
            _lib.check(lib.iisan_gemm16_stream(A.data_ptr(), W.data_ptr(), b.data_ptr(), x.data_ptr(), part.data_ptr(), M, N, K, S, _stream()), "gemm16_stream")
            xg = x.clone()
            _lib.check(lib.iisan_stream_stats_finalize(part.data_ptr(), N // 64, Mp, x.data_ptr(), xc.data_ptr(), rstat.data_ptr(), 1e-6, items, S, _stream()), "finalize")
            runs.append((xg, part, x, xc, rstat))
    finally:
        _lib.dev_set("gemm16_variant", 0)
    torch.cuda.synchronize()
    for r in runs[1:]:
        assert all(torch.equal(a, b2) for a, b2 in zip(r, runs[0]))
    xg, part, x, xc, rstat = runs[0]
    d = A[:M].float() @ W.float().t() + b
    cls = torch.arange(M, device="cuda") % S == 0
    new = torch.where(cls[:, None], d, x0[:M].float() + d)                 # the unrounded sums the kernel forms
    assert (xg[:M].float() - new).abs().max().item() < 0.02                # O(1..20) values: fp16 rounding 8e-3
    assert torch.equal(xg[M:], x0[M:])                                     # rows beyond M untouched
    want_s = new.reshape(M, N // 64, 64).sum(2).t()
    want_q = (new * new).reshape(M, N // 64, 64).sum(2).t()
    assert (part[:, :M, 0] - want_s).abs().max().item() < 2e-3 and ((part[:, :M, 1] - want_q).abs() / want_q).max().item() < 1e-5
    # finalize: CLS rows — fp32 stream += fp16 delta, the stream's slot = the rounded sum; rstd of every (rounded) row
    xc_want = xc0 + xg[:M][cls].float()                                    # the delta exactly as the product left it (held to d above)
    assert torch.equal(xc, xc_want)
    assert torch.equal(x[:M][cls], xc.half())
    assert torch.equal(x[:M][~cls], xg[:M][~cls])
    xr = x[:M].float()
    rs = torch.rsqrt(xr.var(1, unbiased=False) + 1e-6)
    assert ((rstat[:M] - rs).abs() / rs).max().item() < 3e-4                # sums of the unrounded values, single-pass variance


@pytest.mark.parametrize("dt", [0, 1])
def test_layernorm768_vs_torch(lib, dt):
    g = torch.Generator().manual_seed(3)
    rows = 1001
    x = torch.randn(rows, 768, generator=g) * 3 + 0.7
    w = 1 + 0.1 * torch.randn(768, generator=g)
    b = 0.1 * torch.randn(768, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (768,), w.double(), b.double(), 1e-12)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()      # keep the device tensors alive across the launch
    o16 = torch.empty(rows, 768, dtype=T16[dt], device="cuda")
    o32 = torch.empty(rows, 768, dtype=torch.float32, device="cuda")
    _lib.check(lib.iisan_layernorm768(dt, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 1e-12,
                                      o16.data_ptr(), o32.data_ptr(), rows, _stream()), "layernorm768")
    torch.cuda.synchronize()
    assert (o32.cpu().double() - ref).abs().max().item() < 5e-6 * ref.abs().max().item() + 1e-5
    assert (o16.cpu().double() - ref).abs().max().item() < TOL[dt] * ref.abs().max().item()
    # in-place fp32 output (BERT post-LN residual stream)
    _lib.check(lib.iisan_layernorm768(dt, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 1e-12,
                                      None, xd.data_ptr(), rows, _stream()), "layernorm768 in place")
    torch.cuda.synchronize()
    assert (xd.cpu().double() - ref).abs().max().item() < 5e-6 * ref.abs().max().item() + 1e-5


def _attn_ref(qkv, key_bias, items, S, heads):
    D = heads * 64
    x = qkv.double().view(items, S, 3, heads, 64)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) / 8.0
    if key_bias is not None:
        masked = (key_bias < 0)[:, None, None, :]
        s = torch.where(masked, torch.full_like(s, torch.finfo(torch.float32).min), s)
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(items * S, D)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("case", [(3, 197, 12, False), (5, 30, 12, True), (4, 5, 12, False), (3, 8, 2, True),
                                  (2, 64, 3, False), (2, 100, 2, True),
                                  # 14 key tiles (S = 209..224): ADVICE r4 — the V^T row stride of the 13-tile kernel (212) was applied to
                                  # this instantiation too: neighbouring head dims overlapped and the last row ran into the key limits
                                  (2, 224, 2, False), (2, 209, 3, True), (1, 208, 2, True),
                                  # round 5: every tile count's MASKALL = false instantiation (no key_bias, only the last tile can hold pad slots)
                                  (3, 20, 2, False), (2, 120, 2, False), (2, 128, 3, False), (2, 32, 2, False)])
def test_attention16_vs_torch(lib, dt, case):
    items, S, heads, masked = case
    g = torch.Generator().manual_seed(S * 13 + heads)
    D = heads * 64
    qkv = (torch.randn(items * S, 3 * D, generator=g) * 1.5).to(T16[dt])
    kb = None
    if masked:
        kb = torch.zeros(items, S)
        for i in range(items):
            n = int(torch.randint(1, S + 1, (1,), generator=g))
            kb[i, n:] = -1.0
        kb[0, :] = -1.0                       # an all-masked (padding) item attends uniformly
    ref = _attn_ref(qkv, kb, items, S, heads)
    ctx = torch.empty(items * S, D, dtype=T16[dt], device="cuda")
    kbd = kb.cuda() if kb is not None else None
    # kernel input is head-major [items, heads, 3, S, 64]
    qkvd = qkv.view(items, S, 3, heads, 64).permute(0, 3, 2, 1, 4).contiguous().cuda()
    _lib.check(lib.iisan_attention16(dt, qkvd.data_ptr(), kbd.data_ptr() if kbd is not None else None,
                                     ctx.data_ptr(), items, S, heads, _stream()), "attention16")
    torch.cuda.synchronize()
    err = (ctx.cpu().double() - ref).abs().max().item()
    tol = 2.5 * TOL[dt] * ref.abs().max().item()     # P and O are both rounded to 16 bit
    assert err <= tol, f"attention dt={dt} {case}: max err {err:.3e} > {tol:.3e}"
    # CLS-query-only variant (last live encoder block): row 0 of every item
    ctx_cls = torch.empty(items, D, dtype=T16[dt], device="cuda")
    _lib.check(lib.iisan_attention_cls16(dt, qkvd.data_ptr(), kbd.data_ptr() if kbd is not None else None,
                                         ctx_cls.data_ptr(), items, S, heads, _stream()), "attention_cls16")
    torch.cuda.synchronize()
    ref_cls = ref.view(items, S, D)[:, 0]
    err = (ctx_cls.cpu().double() - ref_cls).abs().max().item()
    assert err <= tol, f"attention_cls dt={dt} {case}: max err {err:.3e} > {tol:.3e}"


def test_attention16_at_the_production_grid_is_bit_reproducible_and_right(lib):
    """Round 5 (the kernel lost its per-tile branches and an inline-asm v_min that was only safe behind them): the ViT production launch —
    1,408 items x 12 heads x 197 tokens, 8,448 workgroups, two per CU with the next head's Q / K / V in flight — twice: bit-equal outputs
    (a scheduling hazard or an LDS race shows up as run-to-run differences at this occupancy, not in a 3-item launch), and items from the
    first, a middle and the last workgroup against fp32 softmax(QK^T / 8) V computed on the device."""
    items, S, heads = 1408, 197, 12
    D = heads * 64
    g = torch.Generator(device="cuda").manual_seed(4)
    qkvd = (torch.randn(items, heads, 3, S, 64, device="cuda", generator=g) * 1.5).half()
    out = []
    for _ in range(2):
        ctx = torch.empty(items * S, D, dtype=torch.float16, device="cuda")
        _lib.check(lib.iisan_attention16(0, qkvd.data_ptr(), None, ctx.data_ptr(), items, S, heads, _stream()), "attention16")
        torch.cuda.synchronize()
        out.append(ctx)
    assert torch.equal(out[0], out[1])
    for it in (0, 1, 703, 1406, 1407):
        q, k, v = (qkvd[it, :, i].float() for i in range(3))                   # [H, S, 64]
        p = torch.softmax(q @ k.transpose(1, 2) / 8.0, dim=-1)
        ref = (p @ v).transpose(0, 1).reshape(S, D)
        got = out[0][it * S:(it + 1) * S].float()
        err = (got - ref).abs().max().item()
        assert err <= 2.5 * TOL[0] * ref.abs().max().item(), (it, err)


X3_CASES = [
    # M, N, K, ta, tb, accumulate, scale_a, scale_b
    (1000, 768, 768, 0, 0, 0, 1.0, 0.03),          # fc_* forward of the side network
    (1408, 1024, 8192, 0, 0, 0, 0.25, 0.01),       # Versa dim-align (split-K path)
    (1408, 768, 768, 0, 1, 0, 1e-5, 0.03),         # dX = dY · W, tiny gradients (dynamic scale)
    (64, 768, 1408, 1, 1, 1, 1e-4, 1.0),           # dW += dY^T · X (both operands transposed, atomic accumulate)
    (1024, 8192, 1408, 1, 1, 1, 1e-3, 0.25),       # Versa dPd += dDP^T · tap
    (130, 72, 100, 0, 0, 0, 1.0, 1.0),             # ragged everything
    (8, 8, 4, 0, 0, 0, 300.0, 1e-8),               # tiny, extreme magnitudes
]


@pytest.mark.parametrize("case", X3_CASES)
def test_gemm_x3_matches_fp64(lib, case):
    """Split-operand GEMM (fp32 operands as hi+lo fp16 planes, three MFMA terms, csrc/split.hip) against the product in
    fp64: within a few fp32 ulps of the result scale — the same class of error as an fp32 FMA chain — for every operand
    layout the side network uses (`nn.Linear` forward, dX, dW) and for operand magnitudes far from 1."""
    M, N, K, ta, tb, acc, sa, sb = case
    g = torch.Generator().manual_seed(M * 7 + N)
    A = (torch.randn((K, M) if ta else (M, K), generator=g) * sa).cuda()
    B = (torch.randn((K, N) if tb else (N, K), generator=g) * sb).cuda()
    bias = None if acc else (torch.randn(N, generator=g) * sa * sb).cuda()
    C0 = (torch.randn(M, N, generator=g) * sa * sb * K ** 0.5).cuda() if acc else None
    C = C0.clone() if acc else torch.full((M, N), float("nan"), device="cuda")
    ws = torch.empty(lib.iisan_gemm_x3_ws_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    _lib.check(lib.iisan_gemm_x3(A.data_ptr(), B.data_ptr(), bias.data_ptr() if bias is not None else None, C.data_ptr(),
                                 M, N, K, ta, tb, acc, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "gemm_x3")
    Ad = (A.t() if ta else A).double()
    Bd = (B if tb else B.t()).double()
    ref = Ad @ Bd
    if bias is not None:
        ref = ref + bias.double()[None]
    if acc:
        ref = ref + C0.double()
    assert torch.isfinite(C).all()
    scale = ref.abs().max().item()
    err = (C.double() - ref).abs().max().item() / scale
    # the fp32 product of the same operands, for scale: the split GEMM must be in the same class
    f32 = (A.t() if ta else A) @ (B if tb else B.t())
    if bias is not None:
        f32 = f32 + bias[None]
    if acc:
        f32 = f32 + C0
    err32 = (f32.double() - ref).abs().max().item() / scale
    assert err < max(4e-6, 8 * err32), (case, err, err32)


@pytest.mark.parametrize("which", ["A", "B"])
def test_gemm_x3_drops_the_zero_lo_plane_of_an_fp16_exact_operand(lib, which):
    """Taps cached in fp16 are exact in fp16: their lo plane is all zeros, a device-side flag says so and the GEMM skips the
    last third of K (where that plane's term sits).  Same result as with a genuinely fp32 operand path, to fp32 roundoff."""
    M, N, K = 1408, 1024, 2048
    g = torch.Generator().manual_seed(9)
    A = torch.randn(M, K, generator=g) * 0.25
    B = torch.randn(N, K, generator=g) * 0.02
    if which == "A":
        A = A.half().float()
    else:
        B = B.half().float()
    A, B = A.cuda(), B.cuda()
    C = torch.full((M, N), float("nan"), device="cuda")
    ws = torch.empty(lib.iisan_gemm_x3_ws_bytes(M, N, K), dtype=torch.uint8, device="cuda")
    _lib.check(lib.iisan_gemm_x3(A.data_ptr(), B.data_ptr(), None, C.data_ptr(), M, N, K, 0, 0, 0, ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream().cuda_stream), "gemm_x3")
    ref = A.double() @ B.double().t()
    err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 4e-6, err
