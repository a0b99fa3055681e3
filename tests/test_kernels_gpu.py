"""Operator-level parity of the HIP kernels (through the C ABI) against plain fp32 PyTorch on the
CPU.  Tolerances are fp32-rounding sized: the kernels compute exact fp32 products with fp32
accumulation, only the summation order differs from the CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from smoltts_amd import engine

    engine.load_library()
    assert torch.cuda.is_available()
    return engine


@pytest.fixture(scope="module")
def ops(E):
    from smoltts_amd import ops

    return ops


def bf16r(t):
    return t.to(torch.bfloat16).float()


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rms_norm_ref(x, g, eps):
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * g


@pytest.mark.parametrize("M", [1, 5, 16, 17, 32, 33, 64, 100])
@pytest.mark.parametrize("K,N", [(768, 2368), (576, 960), (3072, 768), (96, 48)])
def test_gemm_bf16_rms_store(E, ops, M, K, N):
    g = torch.Generator().manual_seed(M * 1000 + K + N)
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    ref = rms_norm_ref(x, gamma, 1e-5) @ w.T
    out = ops.linear(x.cuda(), ops.pack_weight(w), N, prologue=E.PRO_RMSNORM, epilogue=E.EPI_STORE, gamma=gamma.cuda(), eps=1e-5)
    assert rel_err(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("M", [1, 32, 40])
def test_gemm_bf16_resid_inplace_and_bias(E, ops, M):
    g = torch.Generator().manual_seed(M)
    K, N = 1536, 576
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    r = torch.randn(M, N, generator=g)
    ref = r + x @ w.T
    rd = r.cuda()
    out = ops.linear(x.cuda(), ops.pack_weight(w), N, epilogue=E.EPI_RESID, resid=rd, out=rd)
    assert out.data_ptr() == rd.data_ptr()
    assert rel_err(rd.cpu(), ref) < 2e-5
    b = torch.randn(N, generator=g)
    out = ops.linear(x.cuda(), ops.pack_weight(w), N, epilogue=E.EPI_STORE, bias=b.cuda())
    assert rel_err(out.cpu(), x @ w.T + b) < 2e-5


@pytest.mark.parametrize("M", [1, 32])
def test_gemm_bf16_swiglu(E, ops, M):
    g = torch.Generator().manual_seed(3 + M)
    K, I = 768, 3072
    x = torch.randn(M, K, generator=g)
    w1 = bf16r(torch.randn(I, K, generator=g) * 0.04)
    w3 = bf16r(torch.randn(I, K, generator=g) * 0.04)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    n = rms_norm_ref(x, gamma, 1e-5)
    ref = F.silu(n @ w1.T) * (n @ w3.T)
    w13 = torch.stack([w1, w3], dim=1).reshape(2 * I, K)
    out = ops.linear(x.cuda(), ops.pack_weight(w13), 2 * I, prologue=E.PRO_RMSNORM, epilogue=E.EPI_SWIGLU, gamma=gamma.cuda())
    assert out.shape == (M, I)
    assert rel_err(out.cpu(), ref) < 3e-5


def _rope_ref(x, cs):  # x (..., H, 64), cs (..., 1, 32, 2)
    xs = x.reshape(*x.shape[:-1], -1, 2)
    o = torch.stack([xs[..., 0] * cs[..., 0] - xs[..., 1] * cs[..., 1], xs[..., 1] * cs[..., 0] + xs[..., 0] * cs[..., 1]], -1)
    return o.flatten(-2)


@pytest.mark.parametrize("M,Hq,Hkv", [(1, 9, 3), (32, 12, 4), (37, 6, 2)])
def test_gemm_qkv_rope_scatter(E, ops, M, Hq, Hkv):
    from smoltts_amd.packing import rope_table

    g = torch.Generator().manual_seed(M + Hq)
    K = Hq * 64
    N = (Hq + 2 * Hkv) * 64
    slots, cache_len = 5, 40
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.04)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    rope = rope_table(cache_len, 64, 100000.0, bf16=True)
    row_pos = torch.randint(0, cache_len, (M,), generator=g, dtype=torch.int32)
    row_slot = torch.randint(0, slots, (M,), generator=g, dtype=torch.int32)
    # make (slot, pos) pairs unique so the scatter has no write conflicts
    pairs = torch.randperm(slots * cache_len, generator=g)[:M]
    row_slot, row_pos = (pairs // cache_len).int(), (pairs % cache_len).int()
    qkv = rms_norm_ref(x, gamma, 1e-5) @ w.T
    q, k, v = qkv.split([Hq * 64, Hkv * 64, Hkv * 64], dim=-1)
    cs = rope[row_pos.long()][:, None]
    q_ref = _rope_ref(q.view(M, Hq, 64), cs).reshape(M, -1)
    k_ref = _rope_ref(k.view(M, Hkv, 64), cs)
    kc = torch.zeros(slots, Hkv, cache_len, 64).cuda()
    vc = torch.zeros(slots, Hkv, cache_len, 64).cuda()
    out = ops.linear(x.cuda(), ops.pack_weight(w), N, prologue=E.PRO_RMSNORM, epilogue=E.EPI_QKV_ROPE, gamma=gamma.cuda(),
                     rope=rope.cuda(), row_pos=row_pos.cuda(), row_slot=row_slot.cuda(), k_cache=kc, v_cache=vc,
                     n_q_heads=Hq, n_kv_heads=Hkv, cache_len=cache_len)
    assert rel_err(out.cpu(), q_ref) < 3e-5
    kc, vc = kc.cpu(), vc.cpu()
    for m in range(M):
        s, p = int(row_slot[m]), int(row_pos[m])
        assert rel_err(kc[s, :, p], k_ref[m]) < 3e-5
        assert rel_err(vc[s, :, p], v.view(M, Hkv, 64)[m]) < 3e-5
    # untouched cache entries stay zero
    mask = torch.ones(slots, cache_len, dtype=torch.bool)
    mask[row_slot.long(), row_pos.long()] = False
    assert float(kc.permute(0, 2, 1, 3)[mask].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,cin,cout,k", [(1, 2, 512, 1024, 7), (3, 16, 512, 256, 3), (2, 40, 64, 32, 1), (2, 1920, 64, 1, 3)])
def test_conv1d_as_gemm(E, ops, B, T, cin, cout, k):
    from smoltts_amd.packing import conv_as_gemm

    g = torch.Generator().manual_seed(cin + k)
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    ref = F.conv1d(F.pad(F.elu(x), (k - 1, 0)), w, b).transpose(1, 2)  # B,T,cout
    halo = k - 1
    buf = torch.zeros(B, halo + T, cin)
    buf[:, halo:] = x.transpose(1, 2)
    gw, gb = conv_as_gemm(w, b, False, 1)
    out = torch.zeros(B, T, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), cout, w_fp32=True, prologue=E.PRO_ELU, epilogue=E.EPI_STORE,
               bias=gb.cuda(), out=out, M=B * T, K=k * cin, ldx=cin, x_bstride=(halo + T) * cin, rows_per_batch=T,
               ldo=cout, o_bstride=T * cout)
    assert rel_err(out.cpu(), ref) < 3e-5


@pytest.mark.parametrize("B,T,cin,cout,s", [(1, 2, 1024, 512, 8), (2, 16, 512, 256, 6), (3, 7, 128, 64, 4)])
def test_convtranspose1d_as_gemm(E, ops, B, T, cin, cout, s):
    from smoltts_amd.packing import conv_as_gemm

    g = torch.Generator().manual_seed(cin + s)
    k = 2 * s
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cin, cout, k, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose1d(F.elu(x), w, b, stride=s)
    ref = y[..., : y.shape[-1] - (k - s)].transpose(1, 2)  # B, T*s, cout
    buf = torch.zeros(B, 1 + T, cin)
    buf[:, 1:] = x.transpose(1, 2)
    gw, gb = conv_as_gemm(w, b, True, s)
    out = torch.zeros(B, T * s, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), s * cout, w_fp32=True, prologue=E.PRO_ELU, epilogue=E.EPI_STORE,
               bias=gb.cuda(), out=out, M=B * T, K=2 * cin, ldx=cin, x_bstride=(1 + T) * cin, rows_per_batch=T,
               ldo=s * cout, o_bstride=T * s * cout)
    assert rel_err(out.cpu(), ref) < 3e-5


def test_gemm_fp32_gelu_scale_resid(E, ops):
    g = torch.Generator().manual_seed(11)
    M, K, N = 24, 512, 2048
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    out = ops.linear(x.cuda(), ops.pack_weight(w, fp32=True), N, w_fp32=True, epilogue=E.EPI_GELU)
    assert rel_err(out.cpu(), F.gelu(x @ w.T)) < 2e-5
    w2 = torch.randn(K, N, generator=g) / math.sqrt(N)
    h = torch.randn(M, N, generator=g)
    r = torch.randn(M, K, generator=g)
    sc = torch.randn(K, generator=g)
    rd = r.cuda()
    ops.linear(h.cuda(), ops.pack_weight(w2, fp32=True), K, w_fp32=True, epilogue=E.EPI_SCALE_RESID, scale=sc.cuda(), resid=rd, out=rd)
    assert rel_err(rd.cpu(), r + sc * (h @ w2.T)) < 2e-5


def test_gemm_rejects_bad_shapes(E, ops):
    x = torch.zeros(4, 40).cuda()
    with pytest.raises(E.SmolttsError):
        ops.linear(x, torch.zeros(1024, dtype=torch.bfloat16).cuda(), 16, K=40)


@pytest.mark.parametrize("Hq,Hkv,window", [(12, 4, 0), (9, 3, 0), (8, 8, 0), (8, 8, 5)])
def test_attention(E, ops, Hq, Hkv, window):
    g = torch.Generator().manual_seed(Hq * 10 + window)
    slots, cache_len, rows = 4, 300, 9
    kc = torch.randn(slots, Hkv, cache_len, 64, generator=g)
    vc = torch.randn(slots, Hkv, cache_len, 64, generator=g)
    q = torch.randn(rows, Hq * 64, generator=g)
    row_pos = torch.tensor([0, 1, 3, 17, 63, 64, 255, 299, 130], dtype=torch.int32)
    row_slot = torch.tensor([0, 1, 2, 3, 0, 1, 2, 3, 0], dtype=torch.int32)
    out = ops.attention(q.cuda(), kc.cuda(), vc.cuda(), row_pos.cuda(), row_slot.cuda(), Hq, window).cpu()
    G = Hq // Hkv
    for r in range(rows):
        p, s = int(row_pos[r]), int(row_slot[r])
        lo = max(0, p + 1 - window) if window else 0
        K = kc[s, :, lo : p + 1].repeat_interleave(G, dim=0)  # Hq, L, 64
        V = vc[s, :, lo : p + 1].repeat_interleave(G, dim=0)
        qq = q[r].view(Hq, 1, 64)
        a = torch.softmax(qq @ K.transpose(1, 2) / 8.0, dim=-1) @ V
        assert rel_err(out[r], a.reshape(-1)) < 2e-5, (r, p)


@pytest.mark.parametrize("mask_mode", [0, 1])
def test_embed(E, ops, mask_mode):
    g = torch.Generator().manual_seed(5)
    dim, cs, ncb = 576, 2048, 8
    te = bf16r(torch.randn(2368, dim, generator=g)).to(torch.bfloat16)
    ce = bf16r(torch.randn(cs * ncb, dim, generator=g)).to(torch.bfloat16)
    cols = torch.zeros(6, 9, dtype=torch.int32)
    cols[:, 0] = torch.tensor([72, 270, 320, 2367, 1000, 319])
    cols[2:, 1:] = torch.randint(0, cs, (4, 8), generator=g, dtype=torch.int32)
    cols[4, 1] = 0  # torch mask rule zeroes this row's code sum although the token is semantic
    x = ops.embed(cols.cuda(), te.cuda(), ce.cuda(), cs, 0, mask_mode).cpu()
    off = torch.arange(0, cs * ncb, cs)
    vq = ce.float()[cols[:, 1:].long() + off].sum(1)
    keep = (cols[:, 1] != 0) if mask_mode == 0 else ((cols[:, 0] >= 320) & (cols[:, 0] <= 2367))
    ref = te.float()[cols[:, 0].long()] + vq * keep[:, None]
    assert rel_err(x, ref) < 1e-6


def test_argmax_first_index_and_margin(E, ops):
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(7, 2368, generator=g)
    logits[1, 100] = logits[1, 2000] = 50.0  # tie -> first index
    logits[2, 2367] = 60.0
    logits[3, 0] = 60.0
    margin = torch.full((7,), float("inf")).cuda()
    ids = ops.argmax(logits.cuda(), margin).cpu()
    assert ids.tolist() == torch.argmax(logits, dim=-1).tolist()
    assert ids[1] == 100
    t2 = torch.topk(logits, 2, dim=-1).values
    assert torch.allclose(margin.cpu(), t2[:, 0] - t2[:, 1])


def test_layernorm(E, ops):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(33, 512, generator=g) * 3 + 1
    w, b = torch.randn(512, generator=g), torch.randn(512, generator=g)
    out = ops.layernorm(x.cuda(), w.cuda(), b.cuda()).cpu()
    assert rel_err(out, F.layer_norm(x, (512,), w, b, 1e-5)) < 1e-5


@pytest.mark.parametrize("B,T,cin,cout,k", [(4, 480, 128, 64, 1), (2, 900, 64, 32, 3), (3, 400, 256, 128, 3), (2, 1100, 32, 64, 1)])
def test_many_row_conv_with_fused_output_activation(E, ops, B, T, cin, cout, k):
    """M >= 1024 takes the LDS-staged many-row kernel; the producer stores ELU(out) (+ the raw copy)."""
    from smoltts_amd.packing import conv_as_gemm

    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    b = torch.randn(cout, generator=g)
    ref = F.conv1d(F.pad(x, (k - 1, 0)), w, b).transpose(1, 2)  # B,T,cout
    halo = k - 1
    buf = torch.zeros(B, halo + T, cin)
    buf[:, halo:] = x.transpose(1, 2)
    gw, gb = conv_as_gemm(w, b, False, 1)
    out = torch.zeros(B, T, cout).cuda()
    raw = torch.zeros(B, T, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), cout, w_fp32=True, epilogue=E.EPI_STORE, bias=gb.cuda(), out=out,
               M=B * T, K=k * cin, ldx=cin, x_bstride=(halo + T) * cin, rows_per_batch=T, ldo=cout, o_bstride=T * cout,
               elu_out=True, raw_out=raw, raw_bstride=T * cout)
    assert rel_err(raw.cpu(), ref) < 3e-5
    assert rel_err(out.cpu(), F.elu(ref)) < 3e-5
    # residual form: out = ELU(resid + conv)
    res = torch.randn(B, T, cout, generator=g)
    out2 = torch.zeros(B, T, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), cout, w_fp32=True, epilogue=E.EPI_RESID, bias=gb.cuda(), out=out2,
               resid=res.cuda(), M=B * T, K=k * cin, ldx=cin, x_bstride=(halo + T) * cin, rows_per_batch=T, ldo=cout,
               o_bstride=T * cout, ldr=cout, r_bstride=T * cout, elu_out=True)
    assert rel_err(out2.cpu(), F.elu(res + ref)) < 3e-5


def test_many_row_convtranspose(E, ops):
    from smoltts_amd.packing import conv_as_gemm

    g = torch.Generator().manual_seed(77)
    B, T, cin, cout, s = 3, 700, 128, 64, 4
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cin, cout, 2 * s, generator=g) / math.sqrt(cin)
    b = torch.randn(cout, generator=g)
    y = F.conv_transpose1d(x, w, b, stride=s)
    ref = y[..., : y.shape[-1] - s].transpose(1, 2)
    buf = torch.zeros(B, 1 + T, cin)
    buf[:, 1:] = x.transpose(1, 2)
    gw, gb = conv_as_gemm(w, b, True, s)
    out = torch.zeros(B, T * s, cout).cuda()
    ops.linear(buf.cuda(), ops.pack_weight(gw, fp32=True), s * cout, w_fp32=True, epilogue=E.EPI_STORE, bias=gb.cuda(), out=out,
               M=B * T, K=2 * cin, ldx=cin, x_bstride=(1 + T) * cin, rows_per_batch=T, ldo=s * cout, o_bstride=T * s * cout)
    assert rel_err(out.cpu(), ref) < 3e-5
