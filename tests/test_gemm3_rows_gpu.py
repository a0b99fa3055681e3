"""The many-row (prefill) variant of the bf16x3 GEMM: same epilogues, M >= 256."""
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


@pytest.mark.parametrize("M,K,N", [(256, 768, 2368), (301, 576, 960), (1000, 384, 48)])
def test_rows_norm_store(E, ops, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g) * 2
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    out = ops.linear3(x3, ops.pack_weight(w), M, N, K, ssq_in=ssq)
    assert rel_err(out.cpu(), rms_norm_ref(x, gamma, 1e-5) @ w.T) < 2e-5


@pytest.mark.parametrize("M,K,N", [(1000, 3072, 768), (257, 1536, 576)])
def test_rows_resid_emit(E, ops, M, K, N):
    g = torch.Generator().manual_seed(M + K)
    h = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.03)
    r = torch.randn(M, N, generator=g)
    ga = 1 + 0.1 * torch.randn(N, generator=g)
    ref = r + h @ w.T
    x3, _, _ = ops.x3_pack(h.cuda())
    rd = r.cuda()
    ea = ops.x3_alloc(M, N)
    ssq = torch.zeros(M, N // 16).cuda()
    ops.linear3(x3, ops.pack_weight(w), M, N, K, epilogue=E.EPI_RESID, resid=rd, out=rd, emit_a=ea, gamma_a=ga.cuda(), ssq_out=ssq)
    out = rd.cpu()
    assert rel_err(out, ref) < 2e-5
    assert rel_err(ops.x3_to_float(ea, M, N), out * ga) < 1e-6
    assert torch.allclose(ssq.cpu().sum(-1), (out * out).sum(-1), rtol=1e-5)


def test_rows_swiglu_and_qkv(E, ops):
    from smoltts_amd.packing import rope_table

    g = torch.Generator().manual_seed(9)
    M, K, I = 520, 768, 3072
    x = torch.randn(M, K, generator=g)
    gamma = 1 + 0.1 * torch.randn(K, generator=g)
    w1, w3 = bf16r(torch.randn(I, K, generator=g) * 0.04), bf16r(torch.randn(I, K, generator=g) * 0.04)
    n = rms_norm_ref(x, gamma, 1e-5)
    x3, _, ssq = ops.x3_pack(x.cuda(), gamma.cuda())
    hout = ops.x3_alloc(M, I)
    ops.linear3(x3, ops.pack_weight(torch.stack([w1, w3], dim=1).reshape(2 * I, K)), M, 2 * I, K, epilogue=E.EPI_SWIGLU, ssq_in=ssq, x3_out=hout)
    assert rel_err(ops.x3_to_float(hout, M, I), F.silu(n @ w1.T) * (n @ w3.T)) < 3e-5
    Hq, Hkv, slots, cache_len = 12, 4, 8, 80
    N = (Hq + 2 * Hkv) * 64
    w = bf16r(torch.randn(N, K, generator=g) * 0.04)
    rope = rope_table(cache_len, 64, 100000.0, bf16=True)
    pairs = torch.randperm(slots * cache_len, generator=g)[:M]
    row_slot, row_pos = (pairs // cache_len).int(), (pairs % cache_len).int()
    qkv = n @ w.T
    q, k, v = qkv.split([Hq * 64, Hkv * 64, Hkv * 64], dim=-1)
    cs = rope[row_pos.long()][:, None]

    def rot(t):
        ts = t.reshape(*t.shape[:-1], -1, 2)
        return torch.stack([ts[..., 0] * cs[..., 0] - ts[..., 1] * cs[..., 1], ts[..., 1] * cs[..., 0] + ts[..., 0] * cs[..., 1]], -1).flatten(-2)

    kc, vc = torch.zeros(slots, Hkv, cache_len, 64).cuda(), torch.zeros(slots, Hkv, cache_len, 64).cuda()
    out = ops.linear3(x3, ops.pack_weight(w), M, N, K, epilogue=E.EPI_QKV_ROPE, ssq_in=ssq, rope=rope.cuda(), row_pos=row_pos.cuda(),
                      row_slot=row_slot.cuda(), k_cache=kc, v_cache=vc, n_q_heads=Hq, n_kv_heads=Hkv, cache_len=cache_len)
    assert rel_err(out.cpu(), rot(q.view(M, Hq, 64)).reshape(M, -1)) < 3e-5
    kref = rot(k.view(M, Hkv, 64))
    kcc, vcc = kc.cpu(), vc.cpu()
    assert rel_err(kcc[row_slot.long(), :, row_pos.long()], kref) < 3e-5
    assert rel_err(vcc[row_slot.long(), :, row_pos.long()], v.view(M, Hkv, 64)) < 3e-5


@pytest.mark.parametrize("M", [20, 300])
def test_position_zero_publishes_v_as_attention_output(E, ops, M):
    """v_x3: rows at position 0 attend to one key, so the attention output is V (repeated per query head)."""
    from smoltts_amd.packing import rope_table

    g = torch.Generator().manual_seed(M)
    K, Hq, Hkv = 384, 6, 2
    N = (Hq + 2 * Hkv) * 64
    x = torch.randn(M, K, generator=g)
    w = bf16r(torch.randn(N, K, generator=g) * 0.05)
    x3, _, _ = ops.x3_pack(x.cuda())
    rope = rope_table(8, 64, 100000.0, bf16=True)
    pos = torch.zeros(M, dtype=torch.int32)
    slot = torch.arange(M, dtype=torch.int32)
    kc, vc = torch.zeros(M, Hkv, 8, 64).cuda(), torch.zeros(M, Hkv, 8, 64).cuda()
    vx3 = ops.x3_alloc(M, Hq * 64)
    q = ops.linear3(x3, ops.pack_weight(w), M, N, K, epilogue=E.EPI_QKV_ROPE, rope=rope.cuda(), row_pos=pos.cuda(), row_slot=slot.cuda(),
                    k_cache=kc, v_cache=vc, n_q_heads=Hq, n_kv_heads=Hkv, cache_len=8, v_x3=vx3)
    want = ops.attention(q, kc, vc, pos.cuda(), slot.cuda(), Hq)  # the real attention kernel over the one-entry cache
    got = ops.x3_to_float(vx3, M, Hq * 64)
    assert torch.equal(got, want.cpu())
    assert torch.equal(got.view(M, Hkv, Hq // Hkv, 64)[:, :, 0], vc.cpu()[:, :, 0])
