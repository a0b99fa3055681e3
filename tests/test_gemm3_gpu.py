"""Parity of the bf16x3 (X3-operand) GEMM path of the DualAR transformer against fp32 PyTorch on the
CPU: the operand format itself (hi+mid+lo == fp32 value), the RMSNorm-by-partial-sums scheme and
every fused epilogue, through the C ABI."""
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


def bf16r(t):
    return t.to(torch.bfloat16).float()


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rms_norm_ref(x, g, eps):
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * g


@pytest.mark.parametrize("M,K", [(1, 768), (32, 576), (37, 3072)])
def test_x3_pack_roundtrip(ops, M, K):
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g) * torch.logspace(-3, 3, K)[None]  # wide dynamic range
    gamma = 1 + 0.2 * torch.randn(K, generator=g)
    a, b, ssq = ops.x3_pack(x.cuda(), gamma.cuda(), None, two=True)
    ya, yb = ops.x3_to_float(a, M, K), ops.x3_to_float(b, M, K)
    assert float(((ya - x * gamma).abs() / (x * gamma).abs().clamp_min(1e-30)).max()) <= 2 ** -23
    assert float(((yb - x).abs() / x.abs().clamp_min(1e-30)).max()) <= 2 ** -23
    s = ssq.cpu()
    assert torch.allclose(s[:, 0], (x * x).sum(-1), rtol=1e-5) and float(s[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("M", [1, 16, 32, 33, 100])
@pytest.mark.parametrize("K,N", [(768, 2368), (576, 2048), (384, 48)])
def test_gemm3_norm_store(E, ops, M, K, N):
    g = torch.Generator().manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g) * 2
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    ref = rms_norm_ref(x, gamma, 1e-5) @ w.T
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    out = ops.linear3(x3, ops.pack_weight(w), M, N, K, ssq_in=ssq)
    assert rel_err(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("M,K,N", [(32, 768, 6144), (32, 3072, 768), (7, 576, 960)])
def test_gemm3_streamed_weights_are_the_same_numbers(E, ops, M, K, N):
    """SmolttsGemm3Args.w_stream only changes the cache policy of the weight loads (non-temporal hint): same bits out."""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    wt = ops.pack_weight(w)
    a = ops.linear3(x3, wt, M, N, K, ssq_in=ssq)
    b = ops.linear3(x3, wt, M, N, K, ssq_in=ssq, w_stream=True)
    assert torch.equal(a, b)
    assert rel_err(a.cpu(), rms_norm_ref(x, gamma, 1e-5) @ w.T) < 2e-5


@pytest.mark.parametrize("M,K,N", [(32, 3072, 768), (5, 1536, 576), (40, 768, 768)])
def test_gemm3_resid_emit(E, ops, M, K, N):
    g = torch.Generator().manual_seed(M + K)
    h = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.03)
    r = torch.randn(M, N, generator=g)
    ga, gb = 1 + 0.1 * torch.randn(N, generator=g), 1 + 0.1 * torch.randn(N, generator=g)
    ref = r + h @ w.T
    x3, _, _ = ops.x3_pack(h.cuda())
    rd = r.cuda()
    ea, eb = ops.x3_alloc(M, N), ops.x3_alloc(M, N)
    ssq = torch.zeros(M, N // 16).cuda()
    ops.linear3(x3, ops.pack_weight(w), M, N, K, epilogue=E.EPI_RESID, resid=rd, out=rd, emit_a=ea, gamma_a=ga.cuda(),
                emit_b=eb, gamma_b=gb.cuda(), ssq_out=ssq)
    out = rd.cpu()
    assert rel_err(out, ref) < 2e-5
    assert rel_err(ops.x3_to_float(ea, M, N), out * ga) < 1e-6
    assert rel_err(ops.x3_to_float(eb, M, N), out * gb) < 1e-6
    assert torch.allclose(ssq.cpu().sum(-1), (out * out).sum(-1), rtol=1e-5)
    # the emitted operand + partial sums drive the next normed GEMM
    w2 = bf16r(torch.randn(64, N, generator=g) * 0.05)
    nxt = ops.linear3(ea, ops.pack_weight(w2), M, 64, N, ssq_in=ssq)
    assert rel_err(nxt.cpu(), rms_norm_ref(out, ga, 1e-5) @ w2.T) < 3e-5


@pytest.mark.parametrize("M,K,I", [(32, 768, 3072), (1, 576, 1536), (20, 384, 512)])
def test_gemm3_swiglu(E, ops, M, K, I):
    g = torch.Generator().manual_seed(I + M)
    x = torch.randn(M, K, generator=g)
    w1 = bf16r(torch.randn(I, K, generator=g) * 0.04)
    w3 = bf16r(torch.randn(I, K, generator=g) * 0.04)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    n = rms_norm_ref(x, gamma, 1e-5)
    ref = F.silu(n @ w1.T) * (n @ w3.T)
    w13 = torch.stack([w1, w3], dim=1).reshape(2 * I, K)
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    hout = ops.x3_alloc(M, I)
    ops.linear3(x3, ops.pack_weight(w13), M, 2 * I, K, epilogue=E.EPI_SWIGLU, ssq_in=ssq, x3_out=hout)
    assert rel_err(ops.x3_to_float(hout, M, I), ref) < 3e-5


@pytest.mark.parametrize("M,Hq,Hkv", [(1, 9, 3), (32, 12, 4), (37, 6, 2)])
def test_gemm3_qkv_rope(E, ops, M, Hq, Hkv):
    from smoltts_amd.packing import rope_table

    g = torch.Generator().manual_seed(M + Hq)
    K, N = Hq * 64, (Hq + 2 * Hkv) * 64
    slots, cache_len = 5, 40
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.04)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    rope = rope_table(cache_len, 64, 100000.0, bf16=True)
    pairs = torch.randperm(slots * cache_len, generator=g)[:M]
    row_slot, row_pos = (pairs // cache_len).int(), (pairs % cache_len).int()
    qkv = rms_norm_ref(x, gamma, 1e-5) @ w.T
    q, k, v = qkv.split([Hq * 64, Hkv * 64, Hkv * 64], dim=-1)
    cs = rope[row_pos.long()][:, None]

    def rot(t):
        ts = t.reshape(*t.shape[:-1], -1, 2)
        return torch.stack([ts[..., 0] * cs[..., 0] - ts[..., 1] * cs[..., 1], ts[..., 1] * cs[..., 0] + ts[..., 0] * cs[..., 1]], -1).flatten(-2)

    kc, vc = torch.zeros(slots, Hkv, cache_len, 64).cuda(), torch.zeros(slots, Hkv, cache_len, 64).cuda()
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    out = ops.linear3(x3, ops.pack_weight(w), M, N, K, epilogue=E.EPI_QKV_ROPE, ssq_in=ssq, rope=rope.cuda(),
                      row_pos=row_pos.cuda(), row_slot=row_slot.cuda(), k_cache=kc, v_cache=vc, n_q_heads=Hq, n_kv_heads=Hkv,
                      cache_len=cache_len)
    assert rel_err(out.cpu(), rot(q.view(M, Hq, 64)).reshape(M, -1)) < 3e-5
    kref, vref = rot(k.view(M, Hkv, 64)), v.view(M, Hkv, 64)
    kc, vc = kc.cpu(), vc.cpu()
    for m in range(M):
        s, p = int(row_slot[m]), int(row_pos[m])
        assert rel_err(kc[s, :, p], kref[m]) < 3e-5 and rel_err(vc[s, :, p], vref[m]) < 3e-5


def test_gemm3_bias_store_emit(E, ops):
    g = torch.Generator().manual_seed(4)
    M, K, N = 32, 768, 576
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.04)
    b = torch.randn(N, generator=g)
    gam = 1 + 0.1 * torch.randn(N, generator=g)
    x3, _, _ = ops.x3_pack(x.cuda())
    ea, ssq = ops.x3_alloc(M, N), torch.zeros(M, N // 16).cuda()
    out = ops.linear3(x3, ops.pack_weight(w), M, N, K, bias=b.cuda(), emit_a=ea, gamma_a=gam.cuda(), ssq_out=ssq).cpu()
    assert rel_err(out, x @ w.T + b) < 2e-5
    assert rel_err(ops.x3_to_float(ea, M, N), out * gam) < 1e-6


@pytest.mark.parametrize("Hq,Hkv,cache_len,window", [(12, 4, 8, 0), (9, 3, 8, 0), (8, 8, 16, 5), (6, 2, 300, 0)])
def test_attention_short_and_x3_out(E, ops, Hq, Hkv, cache_len, window):
    """The 8-entry depth-transformer cache takes the barrier-free one-wave kernel; both kernels can
    write the result directly as the wo GEMM's X3 operand."""
    g = torch.Generator().manual_seed(Hq + cache_len)
    slots, rows = 5, 11
    kc = torch.randn(slots, Hkv, cache_len, 64, generator=g)
    vc = torch.randn(slots, Hkv, cache_len, 64, generator=g)
    q = torch.randn(rows, Hq * 64, generator=g)
    row_pos = torch.randint(0, cache_len, (rows,), generator=g, dtype=torch.int32)
    row_pos[0], row_pos[1] = 0, cache_len - 1
    row_slot = torch.randint(0, slots, (rows,), generator=g, dtype=torch.int32)
    x3 = ops.x3_alloc(rows, Hq * 64)
    out = ops.attention(q.cuda(), kc.cuda(), vc.cuda(), row_pos.cuda(), row_slot.cuda(), Hq, window, out_x3=x3).cpu()
    G = Hq // Hkv
    ref = torch.zeros_like(out)
    for r in range(rows):
        p, s = int(row_pos[r]), int(row_slot[r])
        lo = max(0, p + 1 - window) if window else 0
        K = kc[s, :, lo: p + 1].repeat_interleave(G, dim=0)
        V = vc[s, :, lo: p + 1].repeat_interleave(G, dim=0)
        ref[r] = (torch.softmax(q[r].view(Hq, 1, 64) @ K.transpose(1, 2) / 8.0, dim=-1) @ V).reshape(-1)
    assert rel_err(out, ref) < 2e-5
    assert rel_err(ops.x3_to_float(x3, rows, Hq * 64), out) < 1e-6


@pytest.mark.parametrize("M", [5, 16, 32])
@pytest.mark.parametrize("K,N,extra", [(768, 6144, 16), (768, 2048, 16), (3072, 1536, 16), (576, 3072, 16), (576, 2048, 16), (1536, 576, 16)])
def test_gemm3_predicate_free_form_is_the_same_numbers(E, ops, M, K, N, extra):
    """gemm3_kernel<.., NW> (NW = 8 or 6 waves x exactly U chunks, whole groups of T column tiles: no predicates on the operand loads)
    against the general form of the same kernel: the same weights with `extra` more output columns make a tile count that is
    not a multiple of T, which takes the general form; a column's sum does not depend on its neighbours, so the common columns
    must agree bit for bit."""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N + extra, K, generator=g) * 0.05)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    full = ops.linear3(x3, ops.pack_weight(w[:N]), M, N, K, ssq_in=ssq)
    general = ops.linear3(x3, ops.pack_weight(w), M, N + extra, K, ssq_in=ssq)
    assert torch.equal(full, general[:, :N])
    assert rel_err(full.cpu(), rms_norm_ref(x, gamma, 1e-5) @ w[:N].T) < 2e-5
