"""The bf16x3-split many-row GEMM of the Mimi decoder (csrc/gemm_b3.hip) against float64 PyTorch on the CPU.  Both operands
are fp32 split exactly into three bf16 pieces; six of the nine piece products are kept (the rest are < 2^-24 relative), so
the bar is the same fp32-rounding-sized tolerance as for the fp32-MFMA kernels it replaces -- and it must agree with them."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from smoltts_amd import engine

    engine.load_library()
    return engine


@pytest.fixture(scope="module")
def ops(E):
    from smoltts_amd import ops

    return ops


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def test_w3_pieces_sum_exactly():
    from smoltts_amd.packing import split3_bf16

    w = torch.randn(64, 96, generator=torch.Generator().manual_seed(1)) * torch.logspace(-6, 3, 96)
    h, m, l = split3_bf16(w)
    assert torch.equal(h.float() + m.float() + l.float(), w)  # 3 x 8 significand bits cover fp32's 24


@pytest.mark.parametrize("M,N,K", [(256, 64, 32), (300, 68, 96), (2048, 1536, 512), (2048, 512, 2048), (1000, 640, 512), (4100, 256, 256),
                                   (513, 132, 64)])
def test_b3_store_matches_float64_and_the_fp32_kernel(E, ops, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.logspace(-2, 2, K)[None]  # wide dynamic range across k
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    w32 = ops.pack_weight(w, fp32=True)
    out = ops.linear(x.cuda(), w32, N, w_fp32=True, bias=b.cuda(), w3=ops.pack_weight_w3(w)).cpu()
    old = ops.linear(x.cuda(), w32, N, w_fp32=True, bias=b.cuda()).cpu()
    e_new, e_old = rel_err(out, ref), rel_err(old, ref)
    print(f"M={M} N={N} K={K}: rel err vs float64: bf16x3 {e_new:.2e}, fp32 MFMA {e_old:.2e}")
    assert e_new < 2e-6 and e_new < 4 * e_old + 2e-7


@pytest.mark.parametrize("M,N,K", [(2000, 640, 512), (2048, 1536, 512), (2048, 512, 2048), (4100, 256, 256)])
def test_b3_three_products_are_the_2_to_minus_16_grade_form(E, ops, M, N, K):
    """SmolttsGemmArgs.b3_products = 3 (SMOLTTS_MIMI_OPT_PRODUCTS): hi*hi + mid*hi + hi*mid only.  The dropped products are each
    <= 2^-16 of the leading one, so the result equals float64 of the operands truncated to their hi + mid pieces up to those
    terms: checked against float64 at 2e-5 relative (measured ~3e-6; six products: < 2e-6 as above) and against exactly that
    truncated model at fp32-summation level.  Shapes: the generic kernel, the chunk-size Linears (conv_xs), split-K fc2, conv-size M."""
    from smoltts_amd.packing import split3_bf16

    g = torch.Generator().manual_seed(M + N + K + 3)
    x = torch.randn(M, K, generator=g) * torch.logspace(-2, 2, K)[None]
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    six = ops.linear(x.cuda(), w32, N, w_fp32=True, bias=b.cuda(), w3=w3).cpu()
    three = ops.linear(x.cuda(), w32, N, w_fp32=True, bias=b.cuda(), w3=w3, b3_products=3).cpu()
    xh, xm, _ = split3_bf16(x)
    wh, wm, _ = split3_bf16(w)
    model = (xh.double() @ wh.double().T + xm.double() @ wh.double().T + xh.double() @ wm.double().T) + b.double()
    e3, e6, em = rel_err(three, ref), rel_err(six, ref), rel_err(three, model)
    print(f"M={M} N={N} K={K}: rel err vs float64: three products {e3:.2e}, six {e6:.2e}; three vs its own model {em:.2e}")
    assert not torch.equal(three, six)
    assert e6 < 2e-6 and e6 < e3 < 2e-5 and em < 2e-6


@pytest.mark.parametrize("M,N,K", [(2048, 512, 2048), (1000, 132, 1600), (256, 64, 1536)])
def test_b3_split_k_is_deterministic_and_matches_float64(E, ops, M, N, K):
    """Long K over few tiles: four workgroups per tile + a fixed-order reduce pass (the Mimi transformer's fc2 shape first)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.logspace(-2, 2, K)[None]
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    res, sc = torch.randn(M, N, generator=g), torch.randn(N, generator=g)
    ref = res.double() + sc.double() * (x.double() @ w.double().T)
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    ws = torch.empty(4 * M * N, device="cuda")
    kw = dict(w_fp32=True, epilogue=E.EPI_SCALE_RESID, scale=sc.cuda(), resid=res.cuda(), w3=w3)
    one = ops.linear(x.cuda(), w32, N, **kw).cpu()
    a = ops.linear(x.cuda(), w32, N, splitk_ws=ws, **kw).cpu()
    b = ops.linear(x.cuda(), w32, N, splitk_ws=ws, **kw).cpu()
    assert torch.equal(a, b)
    assert rel_err(a, ref) < 2e-6 and rel_err(a, ref) < 4 * rel_err(one, ref) + 2e-7
    small = ops.linear(x.cuda(), w32, N, splitk_ws=ws[: M * N], **kw).cpu()  # workspace too small: the unsplit path, same result as without
    assert torch.equal(small, one)


def test_b3_elu_prologue_and_residual(E, ops):
    """First conv of a resnet block on the raw tensor: ELU on the way in (hardware exponential: ~6e-8 absolute), + residual out."""
    g = torch.Generator().manual_seed(11)
    M, K, N = 4096, 256, 128
    x, w = torch.randn(M, K, generator=g) * 2, torch.randn(N, K, generator=g) / math.sqrt(K)
    res = torch.randn(M, N, generator=g)
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    ref = res.double() + F.elu(x.double()) @ w.double().T
    got = ops.linear(x.cuda(), w32, N, w_fp32=True, prologue=E.PRO_ELU, epilogue=E.EPI_RESID, resid=res.cuda(), w3=w3).cpu()
    assert rel_err(got, ref) < 2e-6


@pytest.mark.parametrize("taps,cin,N,T,B", [(3, 256, 128, 2203, 3), (2, 128, 320, 1301, 2), (7, 64, 64, 3101, 4),
                                             (3, 256, 128, 5500, 3), (2, 256, 640, 3001, 6), (3, 128, 128, 1111, 16),  # these three: conv_xs.hip
                                             # conv_ks.hip (> 256 channels in 128-channel slices): the four SEANet layers at the
                                             # benchmark's chunk (conv0, ConvTranspose 1, res1's k3 conv, ConvTranspose 2), ragged tiles
                                             (7, 512, 1024, 64, 32), (2, 1024, 4096, 64, 32), (3, 512, 256, 512, 32), (2, 512, 1536, 512, 32),
                                             (3, 512, 256, 333, 48), (2, 384, 128, 100, 130)])
def test_b3_conv_windows_over_halo_prefixed_slots(E, ops, taps, cin, N, T, B):
    """Causal conv as a GEMM over overlapping windows (row stride = cin, K = taps * cin) of per-slot buffers with taps - 1 halo
    rows: the kernel visits the taps of a channel slice back to back (a different K order than the fp32 kernel) -- same result.
    With <= 256 channels, N = 128 | 640 and >= 256 tiles of 64 rows the window-stationary kernel (conv_xs.hip) takes the call;
    with more channels (a multiple of 128) and >= 256 workgroups of 64 rows x 128..512 columns the sliced one (conv_ks.hip)."""
    g = torch.Generator().manual_seed(taps * 1000 + cin)
    rows = T + taps - 1
    buf = torch.randn(B, rows, cin, generator=g)
    w = torch.randn(N, taps * cin, generator=g) / math.sqrt(taps * cin)
    b = torch.randn(N, generator=g)
    win = torch.stack([buf[:, t:t + taps].reshape(B, taps * cin) for t in range(T)], dim=1)  # [B][T][taps * cin]
    ref = win.double() @ w.double().T + b.double()
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    out = torch.zeros(B, T, N, device="cuda")
    kw = dict(w_fp32=True, bias=b.cuda(), out=out, M=B * T, K=taps * cin, ldx=cin, x_bstride=rows * cin, rows_per_batch=T, ldo=N, o_bstride=T * N)
    ops.linear(buf.cuda(), w32, N, w3=w3, **kw)
    got = out.cpu().clone()
    out.zero_()
    ops.linear(buf.cuda(), w32, N, **kw)
    old = out.cpu()
    assert not torch.equal(got, old)  # (the shapes are large enough for the bf16x3 kernel: a different summation order)
    print(f"taps={taps} cin={cin}: bf16x3 {rel_err(got, ref):.2e}, fp32 MFMA {rel_err(old, ref):.2e}")
    assert rel_err(got, ref) < 2e-6 and rel_err(got, ref) < 4 * rel_err(old, ref) + 2e-7


@pytest.mark.parametrize("M", [2048, 2333])
def test_transformer_linears_at_chunk_size(E, ops, M):
    """wo (layer scale + residual, N = 512) and fc1 (GELU, N = 2048) of the decoder transformer at the row count of a 32 x 32
    chunk: the row-stationary kernel (conv_xs.hip, Linear mode) against float64."""
    g = torch.Generator().manual_seed(31 + M)
    K = 512
    x = torch.randn(M, K, generator=g)
    for N, epi in ((512, E.EPI_SCALE_RESID), (2048, E.EPI_GELU)):
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        b, res, sc = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(N, generator=g)
        w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
        y = x.double() @ w.double().T + b.double()
        ref = res.double() + sc.double() * y if epi == E.EPI_SCALE_RESID else F.gelu(y)
        kw = dict(w_fp32=True, epilogue=epi, bias=b.cuda())
        if epi == E.EPI_SCALE_RESID:
            kw.update(scale=sc.cuda(), resid=res.cuda())
        got = ops.linear(x.cuda(), w32, N, w3=w3, **kw).cpu()
        old = ops.linear(x.cuda(), w32, N, **kw).cpu()
        assert not torch.equal(got, old)
        assert rel_err(got, ref) < 2e-6 and rel_err(got, ref) < 4 * rel_err(old, ref) + 2e-7


@pytest.mark.parametrize("K,N,T,B", [(128, 256, 1100, 15), (256, 512, 700, 24)])
def test_k1_conv_plus_residual_over_slots(E, ops, K, N, T, B):
    """The conv that ends a resnet block: k = 1 over per-slot rows, + the block input read with its own strides (it sits two halo
    rows into a wider buffer), ELU on the way out -- conv_xs.hip in Linear mode with 64-row tiles."""
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(B, T, K, generator=g)
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    resbuf = torch.randn(B, T + 2, N, generator=g)
    ref = F.elu(resbuf[:, 2:].double() + x.double() @ w.double().T + b.double())
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    out = torch.zeros(B, T, N, device="cuda")
    rb = resbuf.cuda()
    kw = dict(w_fp32=True, epilogue=E.EPI_RESID, bias=b.cuda(), resid=rb[:, 2:], ldr=N, r_bstride=(T + 2) * N, out=out, M=B * T, K=K,
              ldx=K, x_bstride=T * K, rows_per_batch=T, ldo=N, o_bstride=T * N, elu_out=True)
    ops.linear(x.cuda(), w32, N, w3=w3, **kw)
    got = out.cpu().clone()
    out.zero_()
    ops.linear(x.cuda(), w32, N, **kw)
    old = out.cpu()
    assert not torch.equal(got, old)
    assert rel_err(got, ref) < 2e-6 and rel_err(got, ref) < 4 * rel_err(old, ref) + 2e-7


@pytest.mark.parametrize("M", [2, 7, 33, 70, 600, 2048, 2077])
def test_layernorm_prologue(E, ops, M):
    """nn.LayerNorm as the GEMM's prologue: fused into the skinny kernel (few rows), into the row-stationary kernel (>= 2048 rows,
    K = 512) or applied by the stand-alone kernel into the scratch first (the shapes in between) -- fc1 (GELU) and wqkv (RoPE +
    cache scatter) of the decoder transformer, against float64 / against the unfused pair of launches."""
    g = torch.Generator().manual_seed(77 + M)
    K, N = 512, 2048
    x = torch.randn(M, K, generator=g) * torch.logspace(-1, 1, K)[None] + 0.3
    lw, lb = torch.randn(K, generator=g), torch.randn(K, generator=g)
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    ref = F.gelu(F.layer_norm(x.double(), (K,), lw.double(), lb.double(), 1e-5) @ w.double().T + b.double())
    scratch = torch.zeros(M, K, device="cuda")
    got = ops.linear(x.cuda(), w32, N, w_fp32=True, prologue=E.PRO_LAYERNORM, gamma=lw.cuda(), beta=lb.cuda(), eps=1e-5, ln_scratch=scratch,
                     epilogue=E.EPI_GELU, bias=b.cuda(), w3=w3).cpu()
    two = ops.linear(ops.layernorm(x.cuda(), lw.cuda(), lb.cuda()), w32, N, w_fp32=True, epilogue=E.EPI_GELU, bias=b.cuda(), w3=w3).cpu()
    print(f"M={M}: fused {rel_err(got, ref):.2e}, two launches {rel_err(two, ref):.2e}")
    assert rel_err(got, ref) < 3e-6 and rel_err(got, ref) < 4 * rel_err(two, ref) + 3e-7
    # wqkv: q rows and the K / V scatter equal the unfused pair's within fp32 rounding
    H, cache_len, slots = 8, 64, (M + 63) // 64
    Nq = 3 * H * 64
    wq = torch.randn(Nq, K, generator=g) / math.sqrt(K)
    wq32, wq3 = ops.pack_weight(wq, fp32=True), ops.pack_weight_w3(wq)
    pairs = torch.randperm(slots * cache_len, generator=g)[:M]
    row_slot, row_pos = (pairs // cache_len).int().cuda(), (pairs % cache_len).int().cuda()
    ang = torch.outer(torch.arange(cache_len).float(), 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))
    rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous().cuda()
    outs = []
    for fused in (True, False):
        kc, vc = torch.zeros(slots, H, cache_len, 64).cuda(), torch.zeros(slots, H, cache_len, 64).cuda()
        kw = dict(w_fp32=True, epilogue=E.EPI_QKV_ROPE, rope=rope, row_pos=row_pos, row_slot=row_slot, k_cache=kc, v_cache=vc,
                  n_q_heads=H, n_kv_heads=H, cache_len=cache_len, w3=wq3)
        if fused:
            q = ops.linear(x.cuda(), wq32, Nq, prologue=E.PRO_LAYERNORM, gamma=lw.cuda(), beta=lb.cuda(), eps=1e-5, ln_scratch=scratch, **kw)
        else:
            q = ops.linear(ops.layernorm(x.cuda(), lw.cuda(), lb.cuda()), wq32, Nq, **kw)
        outs.append((q.cpu(), kc.cpu(), vc.cpu()))
    for a_, b_ in zip(*outs):
        assert rel_err(a_, b_) < 3e-6


def test_b3_epilogues(E, ops):
    g = torch.Generator().manual_seed(5)
    M, K, N = 1024, 512, 512
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    y = (x.double() @ w.double().T).float()
    got = ops.linear(x.cuda(), w32, N, w_fp32=True, epilogue=E.EPI_GELU, w3=w3).cpu()
    assert rel_err(got, F.gelu(y)) < 3e-6
    res, sc = torch.randn(M, N, generator=g), torch.randn(N, generator=g)
    got = ops.linear(x.cuda(), w32, N, w_fp32=True, epilogue=E.EPI_SCALE_RESID, scale=sc.cuda(), resid=res.cuda(), w3=w3).cpu()
    assert rel_err(got, res + sc * y) < 3e-6
    got = ops.linear(x.cuda(), w32, N, w_fp32=True, epilogue=E.EPI_RESID, resid=res.cuda(), elu_out=True, w3=w3).cpu()
    assert rel_err(got, F.elu(res + y)) < 3e-6
    raw = torch.zeros(M, N).cuda()
    got = ops.linear(x.cuda(), w32, N, w_fp32=True, epilogue=E.EPI_STORE, elu_out=True, raw_out=raw, w3=w3).cpu()
    assert rel_err(raw.cpu(), y) < 3e-6 and rel_err(got, F.elu(y)) < 3e-6


@pytest.mark.parametrize("M", [600, 2048, 2101])  # >= 2048 rows: the row-stationary kernel (conv_xs.hip, Linear mode)
def test_b3_qkv_rope_equals_the_fp32_kernel_layout(E, ops, M):
    """Same epilogue code as the fp32 many-row kernel: q rows, K/V cache scatter, interleaved-pair rotation."""
    g = torch.Generator().manual_seed(8)
    K, H = 512, 8
    N = 3 * H * 64
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    w32, w3 = ops.pack_weight(w, fp32=True), ops.pack_weight_w3(w)
    slots, cache_len = 9, 256
    pairs = torch.randperm(slots * cache_len, generator=g)[:M]
    row_slot, row_pos = (pairs // cache_len).int().cuda(), (pairs % cache_len).int().cuda()
    ang = torch.outer(torch.arange(cache_len).float(), 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)))
    rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous().cuda()
    outs = []
    for kw in ({}, {"w3": w3}):
        kc, vc = torch.zeros(slots, H, cache_len, 64).cuda(), torch.zeros(slots, H, cache_len, 64).cuda()
        q = ops.linear(x.cuda(), w32, N, w_fp32=True, epilogue=E.EPI_QKV_ROPE, rope=rope, row_pos=row_pos, row_slot=row_slot,
                       k_cache=kc, v_cache=vc, n_q_heads=H, n_kv_heads=H, cache_len=cache_len, **kw)
        outs.append((q.cpu(), kc.cpu(), vc.cpu()))
    for a, b in zip(*outs):
        assert rel_err(b, a) < 3e-6


def test_b3_conv_and_convtranspose_windows(E, ops):
    """Convolutions as GEMMs over halo-prefixed channel-last rows (per-slot strides, rows_per_batch), fused ELU + raw copy."""
    from smoltts_amd.packing import conv_as_gemm

    g = torch.Generator().manual_seed(21)
    B, T, cin, cout, k = 3, 400, 256, 128, 3
    x = torch.randn(B, cin, T, generator=g)
    w, b = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k), torch.randn(cout, generator=g)
    ref = F.conv1d(F.pad(x, (k - 1, 0)).double(), w.double(), b.double()).transpose(1, 2)
    buf = torch.zeros(B, k - 1 + T, cin)
    buf[:, k - 1:] = x.transpose(1, 2)
    gw, gb = conv_as_gemm(w, b, False, 1)
    out, raw = torch.zeros(B, T, cout).cuda(), torch.zeros(B, T, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), cout, w_fp32=True, epilogue=E.EPI_STORE, bias=gb.cuda(), out=out, M=B * T,
               K=k * cin, ldx=cin, x_bstride=(k - 1 + T) * cin, rows_per_batch=T, ldo=cout, o_bstride=T * cout, elu_out=True,
               raw_out=raw, raw_bstride=T * cout, w3=ops.pack_weight_w3(gw))
    assert rel_err(raw.cpu(), ref) < 3e-6 and rel_err(out.cpu(), F.elu(ref)) < 3e-6
    B, T, cin, cout, s = 3, 700, 128, 64, 4
    x = torch.randn(B, cin, T, generator=g)
    w, b = torch.randn(cin, cout, 2 * s, generator=g) / math.sqrt(cin), torch.randn(cout, generator=g)
    y = F.conv_transpose1d(x.double(), w.double(), b.double(), stride=s)
    ref = y[..., : y.shape[-1] - s].transpose(1, 2)
    buf = torch.zeros(B, 1 + T, cin)
    buf[:, 1:] = x.transpose(1, 2)
    gw, gb = conv_as_gemm(w, b, True, s)
    out = torch.zeros(B, T * s, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), s * cout, w_fp32=True, epilogue=E.EPI_STORE, bias=gb.cuda(), out=out, M=B * T,
               K=2 * cin, ldx=cin, x_bstride=(1 + T) * cin, rows_per_batch=T, ldo=s * cout, o_bstride=T * s * cout, w3=ops.pack_weight_w3(gw))
    assert rel_err(out.cpu(), ref) < 3e-6
