"""Per-kernel parity tests: every C-ABI entry point against a torch fp32 CPU restatement of the same op.

Run on the GPU box: python -m pytest tests -m gpu.  bf16 inputs are rounded first and handed to both sides, so the
tolerances below only cover accumulation order and the final rounding of the output dtype (stated per test).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from cm3p_amd import kernels

    return kernels


def _bf(x):
    return x.to(torch.bfloat16)


def _lib_query(name, *args):
    from cm3p_amd import _lib

    return _lib.query(name, *args)


def _err(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def _assert_close(got, want, atol, rtol, what):
    got = got.float().cpu()
    want = want.float().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    bad = (got - want).abs() > atol + rtol * want.abs()
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} mismatches, max|err|={(got - want).abs().max():.4g}, "
            f"first at {idx}: got {got[tuple(idx)]:.6g} want {want[tuple(idx)]:.6g}")


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (256, 384, 192), (200, 72, 96), (1000, 2304, 768), (130, 8, 8)])
def test_gemm_forward_layout(K, M, N, K_):
    """y = x W^T.  Asymmetric integer-valued data makes any row/col or k-order mix-up show as O(1) errors."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randint(-4, 5, (M, K_), generator=g).float()
    w = torch.randint(-4, 5, (N, K_), generator=g).float()
    want = x @ w.t()  # exact in fp32 and in the MFMA (small integers)
    got = K.gemm(_bf(x).to(DEV), _bf(w).to(DEV), M, N, K_, True, True, 1)  # EPI_F32
    _assert_close(got, want, 0, 0, "fwd f32")
    got16 = K.gemm(_bf(x).to(DEV), _bf(w).to(DEV), M, N, K_, True, True, 0)
    _assert_close(got16, want.to(torch.bfloat16), 0, 0, "fwd bf16")
    r = torch.randn(M, N, generator=g)
    gotr = K.gemm(_bf(x).to(DEV), _bf(w).to(DEV), M, N, K_, True, True, 2, resid=r.to(DEV))
    _assert_close(gotr, want + r, 1e-5, 1e-6, "fwd resid")


@pytest.mark.parametrize("T,N,K_", [(128, 128, 128), (256, 192, 64), (200, 72, 96), (1024, 768, 2304)])
def test_gemm_dgrad_layout(K, T, N, K_):
    """dx[T,K] = dy[T,N] W[N,K]: B operand is contraction-strided (transposed LDS reads)."""
    g = torch.Generator().manual_seed(T + N)
    dy = torch.randint(-4, 5, (T, N), generator=g).float()
    w = torch.randint(-4, 5, (N, K_), generator=g).float()
    got = K.gemm(_bf(dy).to(DEV), _bf(w).to(DEV), T, K_, N, True, False, 1)
    _assert_close(got, dy @ w, 0, 0, "dgrad")


@pytest.mark.parametrize("T,N,K_,split", [(128, 128, 128, 1), (512, 192, 64, 1), (777, 72, 96, 3), (4096, 768, 384, 4)])
def test_gemm_wgrad_layout(K, T, N, K_, split):
    """dW[N,K] = dy[T,N]^T x[T,K]: both operands contraction-strided; split-K combine is exact for integers."""
    g = torch.Generator().manual_seed(T + N + 1)
    dy = torch.randint(-3, 4, (T, N), generator=g).float()
    x = torch.randint(-3, 4, (T, K_), generator=g).float()
    got = K.gemm(_bf(dy).to(DEV), _bf(x).to(DEV), N, K_, T, False, False, 1, split_k=split)
    _assert_close(got, dy.t() @ x, 0, 0, "wgrad")


@pytest.mark.parametrize("M,N,K_", [(8192, 2304, 768), (8200, 2312, 128), (16384, 768, 1152)])
def test_gemm256_forward_and_dgrad_layout(K, M, N, K_):
    """Shapes large enough to take the 256 x 256 LDS-DMA kernel (incl. ragged M / N edges): exact integer check."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randint(-3, 4, (M, K_), generator=g).float()
    w = torch.randint(-3, 4, (N, K_), generator=g).float()
    want = x @ w.t()
    _assert_close(K.gemm(_bf(x).to(DEV), _bf(w).to(DEV), M, N, K_, True, True, 1), want, 0, 0, "fwd256 f32")
    _assert_close(K.gemm(_bf(x).to(DEV), _bf(w).to(DEV), M, N, K_, True, True, 0), want.to(torch.bfloat16), 0, 0, "fwd256 bf16")
    r = torch.randn(M, N, generator=g)
    _assert_close(K.gemm(_bf(x).to(DEV), _bf(w).to(DEV), M, N, K_, True, True, 2, resid=r.to(DEV)), want + r, 1e-5, 1e-6, "fwd256 resid")
    if N % 64 == 0:  # dgrad contracts over N
        dy = torch.randint(-3, 4, (M, N), generator=g).float()
        _assert_close(K.linear_dgrad(_bf(dy).to(DEV), _bf(w).to(DEV)), (dy @ w).to(torch.bfloat16), 0, 0, "dgrad256")


@pytest.mark.parametrize("T,N,K_", [(16384, 768, 2304), (32768, 776, 264)])
def test_gemm256_wgrad_layout(K, T, N, K_):
    g = torch.Generator().manual_seed(T + N)
    dy = torch.randint(-2, 3, (T, N), generator=g).float()
    x = torch.randint(-2, 3, (T, K_), generator=g).float()
    got = K.linear_wgrad(_bf(dy).to(DEV), _bf(x).to(DEV))
    _assert_close(got, dy.t() @ x, 0, 0, "wgrad256")


@pytest.mark.parametrize("M,N,K_", [(8192, 3200, 768), (300, 72, 64), (16384, 776, 128)])
def test_gemm_bias_epilogue(K, M, N, K_):
    """CM3P_EPI_F32_BIAS: C = A B^T + bias[n] on the 256 x 256 and the 128 x 128 kernel (integer-valued operands: exact)."""
    from cm3p_amd._lib import EPI_F32_BIAS

    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-2, 3, (M, K_), generator=g).float()
    b = torch.randint(-2, 3, (N, K_), generator=g).float()
    bias = torch.randint(-50, 50, (N,), generator=g).float() / 4
    got = K.gemm(_bf(a).to(DEV), _bf(b).to(DEV), M, N, K_, True, True, EPI_F32_BIAS, resid=bias.to(DEV))
    _assert_close(got, a @ b.t() + bias, 0, 0, "bias epilogue")


def test_gemm_random_tolerance(K):
    """Random normal data: bf16 inputs, fp32 accumulation; error bound 2e-3 relative to sqrt(K)."""
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(512, 768, generator=g))
    w = _bf(torch.randn(1152, 768, generator=g))
    want = x.float() @ w.float().t()
    got = K.gemm(x.to(DEV), w.to(DEV), 512, 1152, 768, True, True, 1)
    _assert_close(got, want, 2e-3 * math.sqrt(768), 0, "random fwd")


def test_gemm_rejects_bad_arguments(K):
    from cm3p_amd._lib import Cm3pHipError

    x = torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV)  # K = 12 is not a multiple of 8
    with pytest.raises(Cm3pHipError):
        K.gemm(x, x, 16, 16, 12, True, True, 0)
    with pytest.raises(Cm3pHipError):
        K.gemm(x.cpu(), x.cpu(), 16, 16, 12, True, True, 0)  # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------ LayerNorm / embedding
@pytest.mark.parametrize("rows,H", [(5, 64), (1000, 768), (37, 256), (4, 2048)])
def test_layernorm_fwd_bwd(K, rows, H):
    g = torch.Generator().manual_seed(rows + H)
    x = torch.randn(rows, H, generator=g) * 2 + 0.5
    w = 1 + 0.2 * torch.randn(H, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = F.layer_norm(xr, (H,), wr, None, 1e-5)
    dy = torch.randn(rows, H, generator=g)
    dres = torch.randn(rows, H, generator=g)
    y.backward(dy)
    y32, y16, mean, rstd = K.layernorm_fwd(x.to(DEV), w.to(DEV), 1e-5, True, True)
    _assert_close(y32, y.detach(), 2e-5, 1e-5, "ln y f32")
    _assert_close(y16, y.detach().to(torch.bfloat16), 1e-4, 8e-3, "ln y bf16")  # one bf16 ulp
    dx32, dx16, dw = K.layernorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, dres.to(DEV), True)
    _assert_close(dx32, xr.grad + dres, 5e-5, 1e-5, "ln dx")
    _assert_close(dw, wr.grad, 2e-4 * math.sqrt(rows), 1e-5, "ln dw")
    # bf16 dy path
    dy16 = _bf(dy)
    xr.grad = None
    wr.grad = None
    F.layer_norm(xr, (H,), wr, None, 1e-5).backward(dy16.float())
    dx32b, _, dwb = K.layernorm_bwd(dy16.to(DEV), x.to(DEV), w.to(DEV), mean, rstd, None, False)
    _assert_close(dx32b, xr.grad, 5e-5, 1e-5, "ln dx (bf16 dy)")
    _assert_close(dwb, wr.grad, 2e-4 * math.sqrt(rows), 1e-5, "ln dw (bf16 dy)")


@pytest.mark.parametrize("impl", ["sorted", "atomic"])
def test_embed_ln_with_audio_override(K, monkeypatch, impl):
    monkeypatch.setenv("CM3P_EMBED_BWD", impl)
    g = torch.Generator().manual_seed(11)
    V, H, B, S = 50, 128, 3, 40
    audio_id = 49
    table = torch.randn(V, H, generator=g)
    table[0] = 0
    w = 1 + 0.1 * torch.randn(H, generator=g)
    ids = torch.randint(0, V - 1, (B, S), generator=g)
    ids[:, 1:7] = audio_id
    ids[2, 20:23] = audio_id
    n_audio = int((ids == audio_id).sum())
    audio = torch.randn(n_audio, H, generator=g)

    tr = table.clone().requires_grad_(True)
    ar = audio.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    emb = F.embedding(ids, tr, padding_idx=0).clone()
    emb[ids == audio_id] = ar
    y = F.layer_norm(emb, (H,), wr, None, 1e-5)
    dy = torch.randn(B, S, H, generator=g)
    y.backward(dy)

    slot, count = K.audio_slots(ids.to(DEV), audio_id)
    want_slot = torch.where(ids.flatten() == audio_id, torch.cumsum((ids.flatten() == audio_id).int(), 0) - 1, -1)
    assert torch.equal(slot.cpu().long(), want_slot.long())  # integer indexing is bit-exact
    assert int(count.item()) == n_audio
    y32, y16, mean, rstd = K.embed_ln_fwd(ids.to(DEV), table.to(DEV), w.to(DEV), 1e-5, slot, audio.to(DEV), want_bf16=True)
    _assert_close(y32.view(B, S, H), y.detach(), 2e-5, 1e-5, "embed ln")
    d_table, d_audio, dw = K.embed_ln_bwd(dy.reshape(-1, H).to(DEV), ids.to(DEV), table.to(DEV), w.to(DEV), mean, rstd, 0, slot,
                                          audio.to(DEV))
    _assert_close(d_table, tr.grad, 1e-4, 1e-5, "embed d_table")
    assert d_table[0].abs().max().item() == 0.0  # padding row gets no gradient
    _assert_close(d_audio, ar.grad, 1e-4, 1e-5, "embed d_audio")
    _assert_close(dw, wr.grad, 1e-3, 1e-5, "embed dw")


@pytest.mark.parametrize("T,V,skew", [(1, 5, 0.0), (63, 7, 0.0), (64, 300, 0.3), (1025, 3167, 0.0), (40000, 300, 0.3), (131072, 3167, 0.4), (70001, 12286, 0.9)])
def test_token_order_is_the_stable_sort_of_the_ids(K, monkeypatch, T, V, skew):
    """cm3p_token_order (a counting sort in six launches) against what defines it: torch.sort(stable=True) of clamp(ids, -1, V) and the
    run numbering by cumsum.  Integer work: bit-exact.  Shapes: one token, less than a wave, exactly one chunk, a block boundary + 1,
    one id holding 30-90 % of the tokens, the C2 token count at the beatmap vocabulary, the largest vocabulary the kernel covers; ids
    outside the table on both sides."""
    g = torch.Generator().manual_seed(T + V)
    ids = torch.randint(0, V, (T,), generator=g)
    if skew:
        ids[torch.rand(T, generator=g) < skew] = V // 3
    if T > 8:
        ids[3], ids[5], ids[T - 1], ids[T // 2] = -7, V, V + 12345, -1
    ids = ids.to(DEV)
    monkeypatch.setenv("CM3P_TOKEN_ORDER", "hip")
    assert K.query("cm3p_token_order_workspace_ints", T, V) > 0
    order, run_of = K.token_order(ids, V, 64)
    monkeypatch.setenv("CM3P_TOKEN_ORDER", "torch")
    order_t, run_of_t = K.token_order(ids, V, 64)
    assert order.dtype == torch.int64 and run_of.dtype == torch.int32
    assert torch.equal(order, order_t) and torch.equal(run_of, run_of_t)


def test_token_order_leaves_large_vocabularies_to_the_torch_route(K):
    assert K.query("cm3p_token_order_workspace_ints", 1000, 12287) == 0  # vocab + 2 > 12288 keys: not covered
    ids = torch.randint(0, 50000, (5000,), device=DEV)
    order, run_of = K.token_order(ids, 50000, 64)
    assert torch.equal(ids[order], torch.sort(ids, stable=True)[0]) and int(run_of[0]) == 0


def test_embedding_backward_in_id_order_is_reproducible_and_matches_autograd(K, monkeypatch):
    """The default embedding backward visits the tokens sorted by id (no atomics): 40000 tokens over a 300-row table with one id
    taking 30 % of them (runs that span many 64-token chunks), the padding row, ids outside the table (no gradient, no fault) and
    audio placeholders.  Against fp32 autograd on the CPU, bit-identical over repeated calls, and equal to the atomic kernel up to
    its summation order."""
    g = torch.Generator().manual_seed(21)
    V, H, T = 300, 768, 40000
    table = torch.randn(V, H, generator=g)
    w = 1 + 0.1 * torch.randn(H, generator=g)
    ids = torch.randint(0, V, (T,), generator=g)
    ids[torch.rand(T, generator=g) < 0.3] = 17
    audio_id = 299
    ids[100:164] = audio_id
    n_audio = int((ids == audio_id).sum())
    audio = torch.randn(n_audio, H, generator=g)
    bad = ids.clone()
    bad[5], bad[6], bad[7] = -3, V, V + 1000  # what nn.Embedding would assert on: zero rows, no gradient
    dy = torch.randn(T, H, generator=g)

    # float64 autograd: row 17 sums 12000 token gradients, where a sequential fp32 sum is itself only good to ~1e-3
    tr, ar, wr = (t.double().requires_grad_(True) for t in (table, audio, w))
    safe = bad.clamp(0, V - 1)
    emb = F.embedding(safe, tr, padding_idx=0).clone()
    outside = (bad < 0) | (bad >= V)
    emb[outside] = 0.0
    emb[bad == audio_id] = ar
    F.layer_norm(emb, (H,), wr, None, 1e-5).backward(dy.double())

    slot, _ = K.audio_slots(bad.to(DEV), audio_id)
    args = (bad.to(DEV), table.to(DEV), w.to(DEV))
    _, _, mean, rstd = K.embed_ln_fwd(*args, 1e-5, slot, audio.to(DEV), want_bf16=False)
    run = lambda: K.embed_ln_bwd(dy.to(DEV), *args, mean, rstd, 0, slot, audio.to(DEV))
    monkeypatch.setenv("CM3P_EMBED_BWD", "sorted")
    a = run()
    b = run()
    assert all(torch.equal(x, y) for x, y in zip(a, b))  # fixed summation order
    _assert_close(a[0], tr.grad, 2e-4, 3e-4, "d_table")
    assert a[0][0].abs().max().item() == 0.0 and a[0][audio_id].abs().max().item() == 0.0
    _assert_close(a[1], ar.grad, 1e-4, 1e-5, "d_audio")
    _assert_close(a[2], wr.grad, 2e-3, 2e-3, "dw")
    monkeypatch.setenv("CM3P_EMBED_BWD", "atomic")
    c = run()
    _assert_close(c[0], tr.grad, 2e-4, 5e-3, "atomic d_table")  # (12000 atomic adds in arrival order on row 17)
    assert torch.equal(a[1], c[1])


# ------------------------------------------------------------------------------------------------ RoPE / GeGLU / pool
@pytest.mark.parametrize("per_batch", [False, True])
def test_rope_matches_reference_formula(K, per_batch):
    from oracle import cm3p_oracle as O

    g = torch.Generator().manual_seed(5)
    B, S, nh = 2, 300, 3
    qkv = _bf(torch.randn(B, S, 3, nh, 64, generator=g))
    pos = torch.arange(S).unsqueeze(0)
    if per_batch:
        pos = torch.stack([torch.arange(S), torch.arange(S) * 3 + 5])
    cos_w, sin_w = O.rope_cos_sin(pos, 160000.0, 64)
    q = qkv[:, :, 0].transpose(1, 2).float()
    k = qkv[:, :, 1].transpose(1, 2).float()
    q2, k2 = O.apply_rope(q, k, cos_w, sin_w)

    inv_freq = O.rope_inv_freq(160000.0, 64)
    cos, sin = K.rope_table(pos.to(DEV), inv_freq.to(DEV))
    _assert_close(cos.view(pos.shape[0], S, 32), cos_w[..., :32], 2e-6, 0, "cos table")
    _assert_close(sin.view(pos.shape[0], S, 32), sin_w[..., :32], 2e-6, 0, "sin table")
    buf = qkv.clone().to(DEV)
    K.rope_apply_(buf, cos, sin, B, S, nh, per_batch)
    _assert_close(buf[:, :, 0].transpose(1, 2), q2.to(torch.bfloat16), 1e-4, 8e-3, "rope q")
    _assert_close(buf[:, :, 1].transpose(1, 2), k2.to(torch.bfloat16), 1e-4, 8e-3, "rope k")
    assert torch.equal(buf[:, :, 2].cpu(), qkv[:, :, 2])  # v untouched
    # inverse rotation is the transpose: rotating back returns the input up to bf16 rounding
    K.rope_apply_(buf, cos, sin, B, S, nh, per_batch, inverse=True)
    _assert_close(buf[:, :, :2], qkv[:, :, :2], 2e-2, 2e-2, "rope inverse")


@pytest.mark.parametrize("T,S,nh,per_batch", [(600, 300, 2, False), (600, 300, 2, True), (8192, 2048, 12, False)])
def test_fused_qkv_rope_gemm_and_inverse_in_attention_backward(K, T, S, nh, per_batch):
    """Wqkv GEMM with the rotary epilogue == plain GEMM followed by the stand-alone RoPE kernel (bf16 ulp: the fused path
    rotates fp32 accumulators before rounding); attention backward with the fused inverse rotation == separate passes."""
    from oracle import cm3p_oracle as O

    g = torch.Generator().manual_seed(T + nh)
    H, B = nh * 64, T // S
    x = _bf(torch.randn(T, H, generator=g)).to(DEV)
    w = _bf(torch.randn(3 * H, H, generator=g) * H ** -0.5).to(DEV)
    pos = torch.arange(S).unsqueeze(0) if not per_batch else torch.stack([torch.arange(S) + 7 * b for b in range(B)])
    cos, sin = K.rope_table(pos.to(DEV), O.rope_inv_freq(10000.0, 64).to(DEV))
    ref = K.linear_fwd(x, w)
    # the packed layout per token is [3][nh][64]; the GEMM output row is exactly that
    K.rope_apply_(ref, cos, sin, B, S, nh, per_batch)
    got = K.qkv_linear_rope(x, w, cos, sin, S, per_batch)
    _assert_close(got, ref, 2e-2, 1.6e-2, "fused qkv rope")
    assert torch.equal(got.view(T, 3, H)[:, 2], ref.view(T, 3, H)[:, 2])  # v third untouched and bit-identical
    # q_scale: only the q third changes - by that factor, applied before the rounding (so within one bf16 ulp of c * q)
    c = K.SOFTMAX_Q_SCALE
    got_s = K.qkv_linear_rope(x, w, cos, sin, S, per_batch, q_scale=c)
    assert torch.equal(got_s.view(T, 3, H)[:, 1:], got.view(T, 3, H)[:, 1:])
    _assert_close(got_s.view(T, 3, H)[:, 0], got.view(T, 3, H)[:, 0].float() * c, 1e-3, 1.0e-2, "q third scaled")

    qkv = got
    do = _bf(torch.randn(T, H, generator=g)).to(DEV)
    out, lse = K.attn_fwd(qkv, None, B, S, nh, -1, 0.125)
    d_sep = K.attn_bwd(qkv, out, do, lse, None, B, S, nh, -1, 0.125)
    K.rope_apply_(d_sep, cos, sin, B, S, nh, per_batch, inverse=True)
    d_fused = K.attn_bwd(qkv, out, do, lse, None, B, S, nh, -1, 0.125, (cos, sin), per_batch)
    scale_ = d_sep.float().abs().max().item()
    _assert_close(d_fused, d_sep, 1e-2 * scale_, 1.6e-2, "fused inverse rope")


def test_geglu_and_gelu(K):
    g = torch.Generator().manual_seed(9)
    T, I = 333, 192
    h = _bf(torch.randn(T, 2 * I, generator=g) * 2)
    hr = h.float().requires_grad_(True)
    a, b = hr.chunk(2, dim=-1)
    y = F.gelu(a) * b
    dg = _bf(torch.randn(T, I, generator=g))
    y.backward(dg.float())
    got = K.geglu_fwd(h.to(DEV))
    _assert_close(got, y.detach().to(torch.bfloat16), 1e-6, 8e-3, "geglu fwd")
    dh = K.geglu_bwd(dg.to(DEV), h.to(DEV))
    _assert_close(dh, hr.grad.to(torch.bfloat16), 1e-5, 8e-3, "geglu bwd")
    x = _bf(torch.randn(64, 128, generator=g) * 3)
    xr = x.float().requires_grad_(True)
    F.gelu(xr).backward(torch.ones(64, 128))
    _assert_close(K.gelu_fwd(x.to(DEV)), F.gelu(x.float()).to(torch.bfloat16), 1e-6, 8e-3, "gelu fwd")
    _assert_close(K.gelu_bwd(torch.ones(64, 128, dtype=torch.bfloat16, device=DEV), x.to(DEV)), xr.grad.to(torch.bfloat16),
                  1e-5, 8e-3, "gelu bwd")


def test_gelu_on_every_finite_bf16_input_against_float64(K):
    """The library's one GELU (csrc/common.h: Phi through the erfcc fit on packed FMAs) on ALL 65280 finite bf16 inputs: the bf16 it
    returns is the correctly rounded fp64 value or its neighbour, also far in the negative tail where 0.5 x (1 + erf) cancels; and the
    derivative gelu'(x) = Phi(x) + x phi(x) likewise."""
    bits = torch.arange(65536, dtype=torch.int32)
    x = (bits << 16).view(torch.float32)
    x = x[torch.isfinite(x)].to(torch.bfloat16)
    assert x.numel() == 65280
    n = x.numel()
    pad = (-n) % 8
    xp = torch.cat([x, torch.zeros(pad, dtype=torch.bfloat16)]).to(DEV)
    y = K.gelu_fwd(xp)[:n].double().cpu()
    dy = K.gelu_bwd(torch.ones_like(xp), xp)[:n].double().cpu()
    xd = x.double()
    cdf = 0.5 * torch.special.erfc(-xd / 2 ** 0.5)
    ref = xd * cdf
    dref = cdf + xd * torch.exp(-0.5 * xd * xd) / (2 * torch.pi) ** 0.5
    big = xd.abs() < 1e30  # (x * Phi(x) overflows bf16 beyond that: inf either way)
    # one bf16 ulp of the exact value (2^-8 relative covers round-to-nearest of a neighbour), plus denormal slack
    assert ((y - ref).abs()[big] <= ref.abs()[big] * 2.0 ** -8 + 1e-37).all(), ((y - ref).abs() / ref.abs().clamp_min(1e-37))[big].max()
    assert ((dy - dref).abs()[big] <= dref.abs()[big] * 2.0 ** -8 + 1e-6).all()
    same = (y.to(torch.bfloat16) == ref.to(torch.bfloat16))[big].float().mean().item()
    assert same > 0.998, same  # (measured: 64 of 65280 round to the neighbouring bf16)


def test_ring_gemm_with_more_workgroups_than_cus_is_the_same_gemm(K, monkeypatch):
    """CM3P_G8P_GRID above the CU count (the switch for sharing the chip with a communication kernel): the surplus
    workgroups walk the same work items - bit-identical results."""
    g = torch.Generator().manual_seed(31)
    M, N, Kd = 17928, 768, 256  # 71 x 3 tiles of 256 x 256 (the ring kernel's threshold is 200), a ragged last row tile
    a, w = _bf(torch.randn(M, Kd, generator=g)).to(DEV), _bf(torch.randn(N, Kd, generator=g) * 0.1).to(DEV)
    r = torch.randn(M, N, generator=g).to(DEV)
    K.gemm8p_set_grid(0)
    want16, want32 = K.linear_fwd(a, w), K.linear_fwd(a, w, resid=r)
    try:
        for grid in (64, 300, 1024):
            K.gemm8p_set_grid(grid)  # (cm3p_gemm8p_set_grid: process-wide, the environment variable is only its initial value)
            assert K.gemm8p_get_grid() == grid
            assert torch.equal(K.linear_fwd(a, w), want16), grid
            assert torch.equal(K.linear_fwd(a, w, resid=r), want32), grid
    finally:
        K.gemm8p_set_grid(0)


@pytest.mark.parametrize("cls,use_mask", [(True, True), (False, True), (False, False)])
def test_pooling(K, cls, use_mask):
    from oracle import cm3p_oracle as O

    g = torch.Generator().manual_seed(13)
    Bn, S, H = 5, 300, 128
    h = torch.randn(Bn, S, H, generator=g)
    lens = torch.tensor([300, 1, 150, 299, 0])
    mask = (torch.arange(S)[None] < lens[:, None]).long() if use_mask else None
    hr = h.clone().requires_grad_(True)
    want = O.pool(hr, mask, cls)
    dp = torch.randn(Bn, H, generator=g)
    want.backward(dp)
    pooled, count = K.pool_fwd(h.to(DEV), mask.to(DEV) if use_mask else None, Bn, S, cls)
    _assert_close(pooled, want.detach(), 1e-5, 1e-5, "pool fwd")
    dh = K.pool_bwd(dp.to(DEV), mask.to(DEV) if use_mask else None, count, Bn, S, cls)
    _assert_close(dh.view(Bn, S, H), hr.grad, 1e-6, 1e-5, "pool bwd")


# ------------------------------------------------------------------------------------------------ attention
def _attn_case(K, B, S, nh, window, lens, seed, check_bwd=True, prescaled=False):
    """prescaled: the kernels are handed q * scale * log2(e) rounded ONCE to bf16 (what the Wqkv GEMM's epilogue produces) and the
    reference runs on exactly that q divided back in fp32 - both modes are held to the same tolerance."""
    from oracle import cm3p_oracle as O

    g = torch.Generator().manual_seed(seed)
    qkv = _bf(torch.randn(B, S, 3, nh, 64, generator=g))
    c = K.SOFTMAX_Q_SCALE
    qkv_dev = qkv.clone()
    if prescaled:
        qkv_dev[:, :, 0] = _bf(qkv[:, :, 0].float() * c)  # one rounding of the scaled q
        qkv = qkv_dev.clone()
        qkv[:, :, 0] = (qkv_dev[:, :, 0].float() / c)  # the reference's q (fp32, not representable in bf16: kept as float below)
    mask = None
    if torch.is_tensor(lens):
        mask = lens.long()  # (an arbitrary (B, S) key mask)
    elif lens is not None:
        mask = (torch.arange(S)[None] < torch.tensor(lens)[:, None]).long()
    allowed = O.attention_allowed(mask, B, S, window if window >= 0 else None)
    if allowed is None and window >= 0:
        allowed = torch.ones(B, 1, S, S, dtype=torch.bool)
    x = qkv.float()
    if prescaled:
        x[:, :, 0] = qkv_dev[:, :, 0].float() / c
    x.requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    o = O.sdpa(q, k, v, allowed, 0.125, eager=True).transpose(1, 2).reshape(B * S, nh * 64)
    do = _bf(torch.randn(B * S, nh * 64, generator=g))
    o.backward(do.float())

    km = mask.to(torch.uint8).to(DEV) if mask is not None else None
    out, lse = K.attn_fwd(qkv_dev.to(DEV), km, B, S, nh, window, 0.125, prescaled=prescaled)
    # 2e-3: q is never re-rounded to bf16 by the kernels (r01 pre-scaled it in bf16 and needed 4e-3)
    _assert_close(out, o.detach().to(torch.bfloat16), 2e-3, 2e-2, f"attn fwd S={S} w={window} prescaled={prescaled}")
    # rows with no visible key are exact zeros
    if allowed is not None:
        dead = ~allowed.any(dim=-1).expand(B, nh, S).transpose(1, 2).reshape(B * S, nh)  # (B*S, nh)
        if dead.any():
            assert out.view(B * S, nh, 64).cpu()[dead].abs().max().item() == 0.0
            assert torch.isinf(lse.cpu().transpose(1, 2).reshape(B * S, nh)[dead]).all()
    if check_bwd:
        dqkv = K.attn_bwd(qkv_dev.to(DEV), out, do.to(DEV), lse, km, B, S, nh, window, 0.125, prescaled=prescaled)
        want = x.grad  # (dq is the gradient w.r.t. the un-scaled q in both modes)
        scale = want.abs().max().item()
        _assert_close(dqkv, want, 2e-2 * scale, 3e-2, f"attn bwd S={S} w={window}")
        for i, nm in enumerate("qkv"):
            e = (dqkv[:, :, i].float().cpu() - want[:, :, i]).norm() / want[:, :, i].norm().clamp_min(1e-9)
            assert e < 2e-2, f"d{nm} relative L2 error {e:.3e}"


def test_attention_global_nopad(K):
    _attn_case(K, 2, 256, 2, -1, None, 1)


@pytest.mark.parametrize("impl", ["fused", "pair"])
@pytest.mark.parametrize("S,lens,prescaled", [(700, [700, 513, 64], True), (1536, None, True), (330, [330, 257, 1], False), (64, None, False)])
def test_attention_global_backward_both_implementations(K, monkeypatch, impl, S, lens, prescaled):
    """Global layers have two backward implementations behind one call: the five-product kernel (attention_bwd_fused.hip, the
    default) and the query-parallel + key-parallel pair (attention_bwd.hip, CM3P_ATTN_BWD_FUSED=0).  Both are held to the same
    tolerance against the fp32 reference, over several 256-key blocks, ragged padding and both q modes."""
    monkeypatch.setenv("CM3P_ATTN_BWD_FUSED", "1" if impl == "fused" else "0")
    _attn_case(K, 2 if lens is None else 3, S, 2, -1, lens, 21, prescaled=prescaled)


@pytest.mark.parametrize("seed", list(range(12)))
def test_attention_random_shapes(K, seed):
    """Seeded sweep over odd shapes: lengths that are not multiples of the 32 / 64 / 256-row blocks, single-row sequences, ragged
    padding (including an almost empty row), one to three heads, global and sliding-window layers, both q modes."""
    g = torch.Generator().manual_seed(1000 + seed)
    S = int(torch.randint(1, 700, (1,), generator=g))
    B = int(torch.randint(1, 4, (1,), generator=g))
    nh = int(torch.randint(1, 4, (1,), generator=g))
    window = -1 if seed % 2 == 0 else 64
    lens = None
    if seed % 3 != 0:
        lens = [S] + [int(torch.randint(1, S + 1, (1,), generator=g)) for _ in range(B - 1)]
    _attn_case(K, B, S, nh, window, lens, 2000 + seed, prescaled=seed % 4 < 2)


@pytest.mark.parametrize("seed", list(range(10)))
def test_attention_global_arbitrary_key_masks(K, seed):
    """The key mask of the reference is any (B, S) 0 / 1 tensor (TF:masking_utils.py:168-179 - right padding is only the usual case).
    Seeded sweep over masks with holes, left padding, a row with one visible key, a row with none, and lengths on both sides of the
    pipelined forward's ring (4 tiles of 64 keys) and its two loops (tiles in front of the first invisible key run without masking
    code, tiles behind the last visible one are not visited): forward and backward against the fp32 restatement, dead rows exact."""
    g = torch.Generator().manual_seed(7000 + seed)
    S = [255, 256, 257, 300, 511, 640, 769, 1024, 1100, 1409][seed]
    B, nh = 4, 1 + seed % 3
    m = torch.ones(B, S, dtype=torch.long)
    m[0] = (torch.rand(S, generator=g) < 0.7).long()                         # holes everywhere
    cut = int(torch.randint(1, S, (1,), generator=g))
    m[1, :cut] = 0                                                           # left padding
    if seed % 2:
        m[1, cut + (S - cut) // 2:] = 0                                      # ... and right padding behind it
    m[2] = 0
    if seed % 3:
        m[2, int(torch.randint(0, S, (1,), generator=g))] = 1                # one visible key (seed % 3 == 0: none at all)
    first = int(torch.randint(S // 2, S, (1,), generator=g))
    m[3, first] = 0                                                          # all visible but one key late in the sweep
    _attn_case(K, B, S, nh, -1, m, 7100 + seed, prescaled=seed % 4 != 3)


def test_attention_fused_backward_matches_pair_and_is_deterministic(K, monkeypatch):
    B, S, nh = 3, 1100, 3
    g = torch.Generator().manual_seed(5)
    qkv = _bf(torch.randn(B, S, 3, nh, 64, generator=g)).to(DEV)
    do = _bf(torch.randn(B * S, nh * 64, generator=g)).to(DEV)
    km = (torch.arange(S)[None] < torch.tensor([1100, 777, 300])[:, None]).to(torch.uint8).to(DEV)
    out, lse = K.attn_fwd(qkv, km, B, S, nh, -1, 0.125)
    monkeypatch.setenv("CM3P_ATTN_BWD_FUSED", "1")
    a = K.attn_bwd(qkv, out, do, lse, km, B, S, nh, -1, 0.125)
    a2 = K.attn_bwd(qkv, out, do, lse, km, B, S, nh, -1, 0.125)
    monkeypatch.setenv("CM3P_ATTN_BWD_FUSED", "0")
    b = K.attn_bwd(qkv, out, do, lse, km, B, S, nh, -1, 0.125)
    assert torch.equal(a, a2)  # partial dq slabs are summed in a fixed order
    for i, (nm, tol) in enumerate((("dq", 6e-3), ("dk", 1e-4), ("dv", 1e-6))):  # dq: one extra bf16 rounding per 256-key partial
        x, y = a[:, :, i].float(), b[:, :, i].float()
        assert ((x - y).norm() / y.norm()).item() < tol, nm
    # rows of padded keys: dk = dv = 0 exactly, in both
    dead = (km == 0).view(B, S)
    assert a[:, :, 1:][dead].abs().max().item() == 0.0 and b[:, :, 1:][dead].abs().max().item() == 0.0


def test_attention_fused_backward_masked_keys_with_unbounded_scores(K, monkeypatch):
    """Keys under the padding mask are not bounded by the row maximum (lse covers visible keys only): with large K rows there,
    exp2(s - lse) overflows.  The fused kernel clamps p to [0, 1] and stores such keys as zero rows of its K image, so they
    contribute exact zeros to dq and get dk = dv = 0."""
    monkeypatch.setenv("CM3P_ATTN_BWD_FUSED", "1")
    B, S, nh = 2, 520, 2
    g = torch.Generator().manual_seed(6)
    qkv = torch.randn(B, S, 3, nh, 64, generator=g)
    lens = torch.tensor([520, 301])
    mask = torch.arange(S)[None] < lens[:, None]
    big = qkv.clone()
    big[:, :, 1][~mask] *= 200.0  # masked keys: scores of several thousand
    do = _bf(torch.randn(B * S, nh * 64, generator=g)).to(DEV)
    km = mask.to(torch.uint8).to(DEV)
    res = []
    for t in (qkv, big):
        x = _bf(t).to(DEV)
        out, lse = K.attn_fwd(x, km, B, S, nh, -1, 0.125)
        res.append(K.attn_bwd(x, out, do, lse, km, B, S, nh, -1, 0.125))
    assert torch.isfinite(res[1].float()).all()
    assert torch.equal(res[0], res[1])  # the masked keys' content is irrelevant, bit for bit


@pytest.mark.parametrize("window,lens,S", [(-1, None, 256), (-1, [203, 100, 7], 203), (64, None, 512), (64, [300, 64, 1], 300)])
def test_attention_with_prescaled_q(K, window, lens, S):
    """The model's mode: q carries scale * log2(e) from the Wqkv GEMM's epilogue (one rounding), the kernels skip the scaling."""
    _attn_case(K, 2 if lens is None else 3, S, 2, window, lens, 11, prescaled=True)


def test_attention_global_padded_ragged(K):
    _attn_case(K, 3, 203, 2, -1, [203, 100, 7], 2)


def test_attention_sliding_window(K):
    _attn_case(K, 2, 512, 2, 64, None, 3)


def test_attention_sliding_window_padded_dead_rows(K):
    # row 1 is valid for 100 keys only: queries >= 165 see no key in a +-64 window -> exact zeros
    _attn_case(K, 2, 384, 2, 64, [384, 100], 4)


def test_attention_short_sequence(K):
    _attn_case(K, 2, 48, 1, 64, None, 5)
    _attn_case(K, 1, 1, 1, -1, None, 6)


def test_attention_late_max_spike_forces_rescale(K):
    """The forward keeps a lazily updated running max (rescale only when a tile max exceeds it by > 2^6).  Force the
    rescale branch late in the key sweep: a key far into the sequence aligned with a few queries makes their max jump by
    orders of magnitude at that tile; and scale half of the queries down so their rows never leave the lazy regime."""
    from oracle import cm3p_oracle as O

    g = torch.Generator().manual_seed(77)
    B, S, nh = 1, 1024, 2
    qkv = torch.randn(B, S, 3, nh, 64, generator=g)
    qkv[:, ::2, 0] *= 0.05                       # flat rows: tile maxima stay within the deferral threshold
    for qi, ki in ((100, 700), (101, 960), (900, 3)):
        qkv[0, ki, 1] = 3.0 * qkv[0, qi, 0]      # score ~ 3 * |q|^2 / 8 >> every other score of that row
    qkv = _bf(qkv)
    x = qkv.float().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    o = O.sdpa(q, k, v, None, 0.125, eager=True).transpose(1, 2).reshape(B * S, nh * 64)
    do = _bf(torch.randn(B * S, nh * 64, generator=g))
    o.backward(do.float())
    out, lse = K.attn_fwd(qkv.to(DEV), None, B, S, nh, -1, 0.125)
    _assert_close(out, o.detach().to(torch.bfloat16), 4e-3, 2e-2, "attn fwd spike")
    s_ref = (q @ k.transpose(-1, -2) * 0.125).detach()
    _assert_close(lse, torch.logsumexp(s_ref, dim=-1), 5e-3, 2e-3, "lse spike")
    dqkv = K.attn_bwd(qkv.to(DEV), out, do.to(DEV), lse, None, B, S, nh, -1, 0.125)
    for i, nm in enumerate("qkv"):
        e = (dqkv[:, :, i].float().cpu() - x.grad[:, :, i]).norm() / x.grad[:, :, i].norm()
        assert e < 2e-2, f"d{nm} relative L2 error {e:.3e}"


def test_attention_long_sequence_properties(K):
    """S = 8192 (BASELINE config 4 length): too big for the O(S^2) oracle in a test, so check properties:
    softmax rows are convex combinations (|out| <= max|v|), lse is finite, and a window-64 run at S = 8192 equals a
    window-64 run on an S = 1024 slice for queries far from the cut."""
    g = torch.Generator().manual_seed(8)
    S, nh = 8192, 1
    qkv = _bf(torch.randn(1, S, 3, nh, 64, generator=g)).to(DEV)
    out, lse = K.attn_fwd(qkv, None, 1, S, nh, -1, 0.125)
    assert torch.isfinite(lse).all() and torch.isfinite(out.float()).all()
    assert out.float().abs().max() <= qkv[:, :, 2].float().abs().max() + 1e-2
    out_w, _ = K.attn_fwd(qkv, None, 1, S, nh, 64, 0.125)
    sl = qkv[:, 2048:3072].contiguous()
    out_s, _ = K.attn_fwd(sl, None, 1, 1024, nh, 64, 0.125)
    a = out_w.view(S, 64)[2048 + 64:3072 - 64].float()
    b = out_s.view(1024, 64)[64:-64].float()
    assert (a - b).abs().max().item() == 0.0  # same tiles, same arithmetic -> bit-identical


# ------------------------------------------------------------------------------------------------ fp32 head
def test_head_kernels(K):
    g = torch.Generator().manual_seed(21)
    a = torch.randn(37, 70, generator=g)
    b = torch.randn(53, 70, generator=g)
    got = K.gemm_f32(a.to(DEV), b.to(DEV), 37, 53, 70, (70, 1), (70, 1), alpha=0.5)
    _assert_close(got, 0.5 * a @ b.t(), 1e-5, 1e-5, "gemm_f32 NT")
    at = torch.randn(70, 37, generator=g)
    got = K.gemm_f32(at.to(DEV), b.to(DEV), 37, 53, 70, (1, 37), (70, 1))
    _assert_close(got, at.t() @ b.t(), 1e-5, 1e-5, "gemm_f32 TN")

    x = torch.randn(9, 512, generator=g)
    xr = x.clone().requires_grad_(True)
    from oracle import cm3p_oracle as O

    y = O.l2_normalize(xr)
    dy = torch.randn(9, 512, generator=g)
    y.backward(dy)
    yy, nrm = K.l2norm_fwd(x.to(DEV))
    _assert_close(yy, y.detach(), 1e-6, 1e-5, "l2norm fwd")
    _assert_close(K.l2norm_bwd(dy.to(DEV), yy, nrm), xr.grad, 1e-6, 1e-4, "l2norm bwd")

    classes = torch.tensor([[1, 0, 2], [0, 0, 3], [2, 3, -1], [3, 1, 0]])
    assert K.first_zero_index(classes.to(DEV)).cpu().tolist() == O.true_variation_index(classes).tolist()

    L = torch.randn(12, 12, generator=g) * 3
    Lr = L.clone().requires_grad_(True)
    loss = O.cm3p_loss(Lr)
    loss.backward()
    d = torch.zeros(12, 12, device=DEV)
    tgt = torch.arange(12, device=DEV)
    lr = K.cross_entropy(L.to(DEV), 12, 12, 12, 1, tgt, None, 0.5 / 12, d)
    lc = K.cross_entropy(L.to(DEV), 12, 12, 1, 12, tgt, None, 0.5 / 12, d)
    got_loss = K.sum_f32(lr, 0.5 / 12)
    K.sum_f32(lc, 0.5 / 12, out=got_loss, accumulate=True)
    _assert_close(got_loss, loss.detach().reshape(1), 1e-6, 1e-6, "clip loss")
    _assert_close(d, Lr.grad, 1e-7, 1e-5, "clip dlogits")


@pytest.mark.parametrize("rows,vocab,frac", [(300, 3167, 0.15), (70, 37, 0.5), (129, 1000, 0.0), (64, 3167, 1.0), (130, 9001, 0.3)])
def test_masked_lm_loss_kernels(K, rows, vocab, frac):
    """cm3p_ce_masked_stats / cm3p_ce_masked_dlogits_bf16 / cm3p_inv_valid_count / cm3p_sum_f32 against
    F.cross_entropy(ignore_index=-100) (TF:loss/loss_utils.py:32-46): loss within fp32 summation-order error, the bf16 gradient
    within one bf16 rounding of the fp32 gradient, its fp32 column sums tight, ignored rows and pad columns exactly zero."""
    g = torch.Generator().manual_seed(rows + vocab)
    pitch = (vocab + 7) // 8 * 8
    x = torch.zeros(rows, pitch)
    x[:, :vocab] = torch.randn(rows, vocab, generator=g) * 2
    lab = torch.randint(0, vocab, (rows,), generator=g)
    lab = torch.where(torch.rand(rows, generator=g) < frac, lab, torch.full_like(lab, -100))
    if frac == 1.0:
        lab[0] = vocab - 1  # last live column is a target
    xr = x[:, :vocab].clone().requires_grad_(True)
    n_valid = int((lab != -100).sum())
    want = F.cross_entropy(xr, lab, ignore_index=-100, reduction="sum") / max(n_valid, 1)
    up = 0.37  # incoming loss gradient
    (want * up).backward()
    xd, ld = x.to(DEV), lab.to(DEV)
    inv = K.inv_valid_count(ld, -100)
    _assert_close(inv, torch.tensor([1.0 / max(n_valid, 1)]), 0, 1e-7, "inv count")
    loss_rows, lse_rows = K.ce_masked_stats(xd, vocab, ld, -100)
    got = K.scale_by(K.sum_f32(loss_rows, 1.0), inv)
    _assert_close(got, want.detach().reshape(1), 1e-6, 1e-5, "masked-LM loss")
    assert torch.equal(loss_rows.cpu()[lab == -100], torch.zeros(rows - n_valid))
    dl, colsum = K.ce_masked_dlogits_bf16(xd, vocab, ld, -100, lse_rows, torch.tensor([up], device=DEV), inv)
    assert dl.dtype == torch.bfloat16 and dl.shape == (rows, pitch)
    wantg = torch.zeros(rows, pitch)
    wantg[:, :vocab] = xr.grad
    _assert_close(dl, wantg, 1e-8, 2 ** -8, "masked-LM dlogits (bf16)")
    assert torch.equal(dl.float().cpu()[lab == -100], torch.zeros(rows - n_valid, pitch))
    assert torch.equal(dl.float().cpu()[:, vocab:], torch.zeros(rows, pitch - vocab))
    _assert_close(colsum, wantg.sum(0), 1e-6, 1e-4, "decoder bias gradient")
    # the 1024-thread reductions at a length that exercises the unrolled body, the tail and every wave
    v = torch.randn(131072 + 777, generator=g)
    _assert_close(K.sum_f32(v.to(DEV), 0.5), (0.5 * v.double().sum()).float().reshape(1), 2e-3, 1e-5, "sum_f32 long")
    t = torch.randint(0, 5, (131072 + 777,), generator=g)
    t = torch.where(t == 0, torch.full_like(t, -100), t)
    _assert_close(K.inv_valid_count(t.to(DEV), -100), torch.tensor([1.0 / int((t != -100).sum())]), 0, 1e-7, "inv count long")


# ------------------------------------------------------------------------------------------------ audio front end
def test_audio_conv_frontend_matches_conv1d(K):
    """im2col + GEMM + bias/GELU against F.conv1d on bf16-rounded weights (tolerance: bf16 activations, 1e-2 relative)."""
    from cm3p_amd.audio import audio_frontend

    g = torch.Generator().manual_seed(31)
    B, n_mels, T, C = 2, 16, 96, 64
    x = torch.randn(B, n_mels, T, generator=g)
    w1 = (torch.randn(C, n_mels, 3, generator=g) * 0.2).requires_grad_(True)
    b1 = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    w2 = (torch.randn(C, C, 3, generator=g) * 0.1).requires_grad_(True)
    b2 = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    y = F.gelu(F.conv1d(F.gelu(F.conv1d(x, w1, b1, padding=1)), w2, b2, stride=2, padding=1)).permute(0, 2, 1)
    dy = torch.randn(B, T // 2, C, generator=g)
    y.backward(dy)
    ps = [p.detach().clone().to(DEV).requires_grad_(True) for p in (w1, b1, w2, b2)]
    got = audio_frontend(x.to(DEV), *ps)
    assert got.shape == (B, T // 2, C) and got.dtype == torch.float32
    got.backward(dy.to(DEV))
    rel = lambda a, b: ((a.float().cpu() - b).norm() / b.norm()).item()
    assert rel(got.detach(), y.detach()) <= 1e-2
    for p, r, nm in zip(ps, (w1, b1, w2, b2), ("w1", "b1", "w2", "b2")):
        assert rel(p.grad, r.grad) <= 2e-2, (nm, rel(p.grad, r.grad))


# ------------------------------------------------------------------------------------------------- unpadded attention
@pytest.mark.parametrize("S,lens,nh", [(1100, [1100, 1023, 65, 1], 2), (512, None, 3), (4096 + 77, [4096 + 77, 2049], 1), (200, [200, 1, 199], 2)])
def test_pipelined_global_forward_agrees_with_the_three_wave_kernel(K, monkeypatch, S, lens, nh):
    """Global layers with pre-scaled q have two forward implementations behind one call: the software-pipelined one-wave-per-SIMD kernel
    (attention_fwd.hip, the default) and attn_fwd_kernel (attention.hip, CM3P_ATTN_FWD_IMPL=wave3).  Same mathematics, different order
    of the row sums and a different moment at which the lazily moved reference point moves: outputs agree to two bf16 roundings, lse to
    fp32 rounding, rows without a visible key are exact zeros / +inf in both, and the pipelined kernel is bit-reproducible."""
    g = torch.Generator().manual_seed(S + nh)
    B = 2 if lens is None else len(lens)
    qkv = _bf(torch.randn(B, S, 3, nh, 64, generator=g)).to(DEV)
    qkv[:, :, 0] = (qkv[:, :, 0].float() * K.SOFTMAX_Q_SCALE).to(torch.bfloat16)
    qkv[:, S // 2:, 1] *= 4.0  # late, much larger scores: the reference point moves long after the first tile
    km = None
    if lens is not None:
        km = (torch.arange(S)[None] < torch.tensor(lens)[:, None]).to(torch.uint8).to(DEV)
        km[0, 3] = 0  # a hole inside the valid range
    monkeypatch.setenv("CM3P_ATTN_FWD_IMPL", "wave3")
    o3, l3 = K.attn_fwd(qkv, km, B, S, nh, -1, 0.125, prescaled=True)
    monkeypatch.delenv("CM3P_ATTN_FWD_IMPL")
    o1, l1 = K.attn_fwd(qkv, km, B, S, nh, -1, 0.125, prescaled=True)
    o2, l2 = K.attn_fwd(qkv, km, B, S, nh, -1, 0.125, prescaled=True)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(l1, l2)
    fin = torch.isfinite(l3)
    assert torch.equal(fin, torch.isfinite(l1))
    assert (l1[fin] - l3[fin]).abs().max().item() <= 2e-5
    if (~fin).any():
        assert bool((l1[~fin] == float("inf")).all())
    # two bf16 roundings apart at most.  The kernels round P to bf16 relative to DIFFERENT reference points (the pipelined one moves a row's
    # reference when that row asks, attn_fwd_kernel when any row of the wave does), so a different subset of the p values rounds up:
    # a relative difference of up to ~2^-8 between the fp32 results, i.e. one or two units of the bf16 output
    d = (o1.float() - o3.float()).abs()
    # (absolute part: with a softmax this peaked - the late keys are scaled by 4 - the rounding of p to bf16 alone moves an output by up to
    # 2^-9 * sum p |v| ~ 2e-3 per kernel, measured against float64: max 1.3e-2 - 1.5e-2, mean 4e-4 for BOTH kernels on these inputs)
    assert (d <= 2 * 0.0079 * o3.float().abs() + 8e-3).all(), (d - 2 * 0.0079 * o3.float().abs()).max().item()


@pytest.mark.parametrize("nkb,lens", [(1, None), (2, [512 - 17, 300]), (3, None), (5, [1280 - 100, 1025, 300])])
def test_fused_backward_with_four_key_blocks_per_slab_at_short_sequences(K, monkeypatch, nkb, lens):
    """CM3P_FUSED_SLAB_GROUP=4 (the rule only picks it from 24 key blocks = S > 5888 on, which no kernel test reaches): partial groups,
    a single key block, padded rows - against the fp32 reference and against the group size 2 the same call picks by itself."""
    S = 256 * nkb - (17 if lens is None else 0)
    if lens is not None:
        S = lens[0]
    monkeypatch.setenv("CM3P_ATTN_BWD_FUSED", "1")
    monkeypatch.setenv("CM3P_FUSED_SLAB_GROUP", "4")
    assert _lib_query("cm3p_attn_bwd_fused_slab_group", S) == 4
    _attn_case(K, 2 if lens is None else len(lens), S, 2, -1, lens, 77 + nkb, prescaled=True)
    monkeypatch.setenv("CM3P_FUSED_SLAB_GROUP", "4x")  # not exactly "2" / "4": ignored by the one place that parses it
    assert _lib_query("cm3p_attn_bwd_fused_slab_group", S) == 2


@pytest.mark.parametrize("window,prescaled", [(-1, False), (-1, True), (64, False)])
def test_attention_varlen_equals_padded_on_valid_rows(K, window, prescaled):
    """cm3p_attn_*_varlen on packed sequences == cm3p_attn_* on the right-padded batch, bit for bit, on every valid row
    (masked keys contribute exact zeros, so packing must not change a single bit), forward and backward with the RoPE epilogue."""
    torch.manual_seed(0)
    B, S, nh = 4, 333, 2
    lens = [333, 200, 97, 64]
    qkv = (torch.randn(B, S, 3, nh, 64, device=DEV) * 0.7).bfloat16()
    do = torch.randn(B * S, nh * 64, device=DEV).bfloat16()
    mask = torch.zeros(B, S, dtype=torch.uint8, device=DEV)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    do = do * mask.reshape(B * S, 1).to(do.dtype)  # padded queries carry no gradient (in the model they never reach the loss)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device=DEV, dtype=torch.float32) / 64))
    cos, sin = K.rope_table(torch.arange(S, device=DEV), inv_freq)
    # (prescaled: the global forward runs the pipelined kernel of attention_fwd.hip, whose masked tiles must also contribute exact zeros)
    out, lse = K.attn_fwd(qkv, mask, B, S, nh, window, 0.125, prescaled)
    dqkv = K.attn_bwd(qkv, out, do, lse, mask, B, S, nh, window, 0.125, (cos, sin), False, prescaled)

    idx = torch.nonzero(mask.flatten()).flatten()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=DEV)
    qkv_p = qkv.reshape(B * S, 3, nh, 64)[idx].contiguous()
    do_p = do[idx].contiguous()
    pos = (idx % S).contiguous()
    cos_p, sin_p = K.rope_table(pos, inv_freq)
    out_p, lse_p = K.attn_fwd_varlen(qkv_p, cu, B, max(lens), nh, window, 0.125, prescaled)
    dqkv_p = K.attn_bwd_varlen(qkv_p, out_p, do_p, lse_p, cu, B, max(lens), nh, window, 0.125, (cos_p, sin_p), prescaled)
    torch.cuda.synchronize()
    assert torch.equal(out_p, out[idx])
    lse_rows = lse.permute(1, 0, 2).reshape(nh, B * S)[:, idx]  # [B, nh, S] -> [nh, total]
    assert torch.equal(lse_p, lse_rows)
    assert torch.equal(dqkv_p, dqkv.reshape(B * S, 3, nh, 64)[idx])


@pytest.mark.parametrize("D,S,window,lens", [(16, 200, -1, None), (16, 333, 64, [333, 100]), (32, 130, 64, [130, 40]), (32, 512, -1, [512, 77]),
                                             (64, 257, -1, [257, 200]), (64, 300, 64, None)])
def test_generic_attention_matches_fp32_reference(K, D, S, window, lens):
    """csrc/attention_generic.hip (head sizes 16 / 32; 64 as a cross-check): forward, lse and the three gradients against fp32 autograd
    of softmax(scale q k^T + mask) v with the reference's mask rule (key padding AND |q - k| <= window; padded queries are not masked);
    rows with no visible key give exact zeros and lse = +inf; at head_dim 64 the MFMA kernels must agree with it."""
    g = torch.Generator().manual_seed(D + S)
    nh = 2
    B = len(lens) if lens else 2
    scale = D ** -0.5
    qkv = _bf(torch.randn(B, S, 3, nh, D, generator=g) * 0.8)
    do = _bf(torch.randn(B * S, nh * D, generator=g) * 0.5)
    mask = None
    if lens is not None:
        mask = (torch.arange(S)[None] < torch.tensor(lens)[:, None])
    x = qkv.float().requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))  # (B, nh, S, D)
    sc = (q @ k.transpose(-1, -2)) * scale
    vis = torch.ones(B, 1, S, S, dtype=torch.bool)
    if mask is not None:
        vis = vis & mask[:, None, None, :]
    if window >= 0:
        idx = torch.arange(S)
        vis = vis & ((idx[:, None] - idx[None, :]).abs() <= window)[None, None]
    sc = sc.masked_fill(~vis, float("-inf"))
    dead = ~vis.any(dim=-1)  # (B, 1, S)
    p = torch.softmax(sc, dim=-1).masked_fill(dead[..., None], 0.0)
    o = (p @ v).transpose(1, 2).reshape(B * S, nh * D)
    o.backward(do.float())
    lse_ref = torch.logsumexp(sc, dim=-1)  # -inf on dead rows (the kernels store +inf there)

    km = mask.to(torch.uint8).to(DEV) if mask is not None else None
    out, lse = K.attn_fwd_generic(qkv.to(DEV), km, B, S, nh, D, window, scale)
    _assert_close(out, o.detach().to(torch.bfloat16), 2e-3, 2e-2, f"generic attn fwd D={D}")
    dead_rows = dead.expand(B, nh, S)
    live = ~dead_rows
    assert torch.allclose(lse.cpu()[live], lse_ref[live], atol=2e-3, rtol=1e-4)
    if dead_rows.any():
        assert torch.isinf(lse.cpu()[dead_rows]).all() and (lse.cpu()[dead_rows] > 0).all()
        assert out.view(B, S, nh, D).cpu().permute(0, 2, 1, 3)[dead_rows].abs().max().item() == 0.0
    dqkv = K.attn_bwd_generic(qkv.to(DEV), out, do.to(DEV), lse, km, B, S, nh, D, window, scale)
    want = x.grad
    for i, nm in enumerate("qkv"):
        e = (dqkv[:, :, i].float().cpu() - want[:, :, i]).norm() / want[:, :, i].norm().clamp_min(1e-9)
        assert e < 1e-2, f"generic d{nm} relative L2 error {e:.3e}"
    if mask is not None:  # padded keys receive exact zeros
        assert dqkv[:, :, 1:].float().cpu()[~mask].abs().max().item() == 0.0
    if D == 64:  # the MFMA kernels on the same inputs
        out64, lse64 = K.attn_fwd(qkv.to(DEV), km, B, S, nh, window, scale)
        _assert_close(out64, out, 2e-3, 2e-2, "MFMA vs generic forward")
        d64 = K.attn_bwd(qkv.to(DEV), out64, do.to(DEV), lse64, km, B, S, nh, window, scale)
        assert ((d64.float() - dqkv.float()).norm() / dqkv.float().norm()).item() < 1e-2


@pytest.mark.parametrize("D", [16, 32, 64])
def test_generic_rope_is_the_reference_rotation_and_its_transpose(K, D):
    """cm3p_rope_apply_generic: q * cos + rotate_half(q) * sin in fp32 on the bf16 values (TF:models/modernbert/modeling_modernbert.py:188-219),
    v untouched; inverse = 1 applies the transposed rotation (the backward)."""
    g = torch.Generator().manual_seed(D)
    B, S, nh = 2, 37, 3
    qkv = _bf(torch.randn(B, S, 3, nh, D, generator=g))
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    cos, sin = K.rope_table(torch.arange(S, device=DEV), inv_freq.to(DEV))
    c = torch.cat((cos.cpu(), cos.cpu()), -1)[None, :, None, :]  # (1, S, 1, D)
    s_ = torch.cat((sin.cpu(), sin.cpu()), -1)[None, :, None, :]
    rot = lambda t: torch.cat((-t[..., D // 2:], t[..., :D // 2]), -1)
    x = qkv.float()
    want = x.clone()
    want[:, :, :2] = x[:, :, :2] * c.unsqueeze(2) + rot(x[:, :, :2]) * s_.unsqueeze(2)
    got = K.rope_apply_generic_(qkv.clone().to(DEV), cos, sin, B, S, nh, D, False)
    assert torch.equal(got[:, :, 2].cpu(), qkv[:, :, 2])
    assert (got.float().cpu() - want).abs().max().item() <= 2e-2  # one bf16 rounding of O(1) values
    back = x.clone()
    back[:, :, :2] = x[:, :, :2] * c.unsqueeze(2) - rot(x[:, :, :2]) * s_.unsqueeze(2)
    got_inv = K.rope_apply_generic_(qkv.clone().to(DEV), cos, sin, B, S, nh, D, False, inverse=True)
    assert (got_inv.float().cpu() - back).abs().max().item() <= 2e-2


def test_gather_scatter_rows(K):
    x = torch.randn(37, 64, device=DEV)
    idx = torch.tensor([5, 0, 36, 7, 8], device=DEV)
    g = K.gather_rows(x, idx)
    assert torch.equal(g, x[idx])
    sc = K.scatter_rows(g, idx, 37)
    want = torch.zeros_like(x)
    want[idx] = x[idx]
    assert torch.equal(sc, want)


@pytest.mark.parametrize("rows,cols", [(2304, 768), (768, 1152), (72, 8), (200, 136)])
def test_cast_with_transpose(K, rows, cols):
    """One pass gives the bf16 copy and its transpose, both holding exactly the values of a plain cast."""
    g = torch.Generator().manual_seed(rows + cols)
    w = torch.randn(rows, cols, generator=g).to(DEV)
    y, yt = K.cast_bf16_with_transpose(w)
    assert torch.equal(y, w.to(torch.bfloat16)) and torch.equal(yt, w.to(torch.bfloat16).t().contiguous())


def test_cast_with_transpose_of_many_matrices_in_one_launch(K):
    """cm3p_cast_f32_bf16_t_multi: every matrix gets exactly what the one-matrix kernel gives (edge tiles, one-tile matrices, a repeated
    shape), and the copies do not overlap."""
    g = torch.Generator().manual_seed(3)
    shapes = [(2304, 768), (768, 768), (768, 1152), (72, 8), (200, 136), (2304, 768), (64, 64), (8, 520)]
    ws = [torch.randn(r, c, generator=g).to(DEV) for r, c in shapes]
    got = K.cast_bf16_with_transpose_many(ws)
    assert len(got) == len(ws)
    for w, (y, yt) in zip(ws, got):
        assert torch.equal(y, w.to(torch.bfloat16)) and torch.equal(yt, w.to(torch.bfloat16).t().contiguous())
    assert K.cast_bf16_with_transpose_many([]) == []


@pytest.mark.parametrize("T,N,K_", [(65536, 768, 1152), (32768 + 64, 2304, 768), (512, 256, 128)])
def test_dgrad_through_transposed_weight_equals_strided_dgrad(K, T, N, K_):
    """dx = dy W computed from W^T (contraction index contiguous in both operands) is bit-identical to the k-strided form:
    the same products in the same accumulation order (big shapes run gemm8p.hip, the small one the 128 x 128 kernel)."""
    g = torch.Generator().manual_seed(T + N)
    dy = torch.randn(T, N, generator=g).to(torch.bfloat16).to(DEV)
    w32 = (torch.randn(N, K_, generator=g) * 0.05).to(DEV)
    w, wt = K.cast_bf16_with_transpose(w32)
    a, b = K.linear_dgrad(dy, w), K.linear_dgrad(dy, w, wt)
    assert torch.equal(a, b)
    ref = dy[:256].float() @ w.float()
    assert (b[:256].float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("T,I,Kd", [(32 * 4096, 1152, 768), (8 * 4096 + 8, 1152, 768), (51200, 512, 256), (52000, 96, 192)])
def test_wi_gemm_with_geglu_in_its_store_phase_equals_the_two_kernels(K, T, I, Kd):
    """cm3p_gemm_geglu (forward-only calls): gelu_erf(h) * g computed on the staged bf16 rows of the Wi GEMM, with the weight rows
    interleaved so that a wave holds the h and the g of the same columns.  Same roundings in the same places as cm3p_gemm_bf16
    followed by cm3p_geglu_fwd: bit-identical - full tiles, a ragged last row tile, the metadata tower's width, and a width whose
    last column tile is three quarters empty (2I = 192) with an odd number of k-tiles."""
    assert K.gemm_geglu_supported(T, I, Kd)
    g = torch.Generator().manual_seed(T % 1000 + I)
    x = _bf(torch.randn(T, Kd, generator=g)).to(DEV)
    w = _bf(torch.randn(2 * I, Kd, generator=g) * 0.05).to(DEV)
    want = K.geglu_fwd(K.linear_fwd(x, w))
    got = K.gemm_geglu(x, w.index_select(0, K.geglu_interleave_index(I, w.device)).contiguous())
    assert torch.equal(got, want)

