"""The codec transformer's chunk attention on bf16x3 piece caches (attn_rows3_kernel) and the QKV epilogue that writes them."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from smoltts_amd import engine as E, ops
    E.load_library()
    return E, ops


def test_piece_cache_round_trip_on_cpu_layout(mods):
    """kv3_encode / kv3_decode are inverse and exact (hi + mid + lo == the fp32 value) in both layouts."""
    _, ops = mods
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 70, 64, generator=g) * torch.logspace(-3, 3, 64)
    for is_v in (False, True):
        enc = ops.kv3_encode(x, is_v)
        assert enc.shape == (2, 3, 96 * 384)
        assert torch.equal(ops.kv3_decode(enc, 70, is_v), x)


def softmax_ref(q, k, v, row_pos, row_slot, heads, window):
    rows = q.shape[0]
    out = torch.zeros(rows, heads * 64, dtype=torch.float64)
    for m in range(rows):
        s, pos = int(row_slot[m]), int(row_pos[m])
        lo = max(0, pos + 1 - window) if window > 0 else 0
        for h in range(heads):
            qq = q[m, h * 64:(h + 1) * 64].double() / 8.0
            sc = k[s, h, lo:pos + 1].double() @ qq
            p = torch.softmax(sc, 0)
            out[m, h * 64:(h + 1) * 64] = p @ v[s, h, lo:pos + 1].double()
    return out


@pytest.mark.parametrize("slots,heads,rps,cache_len,starts,window", [
    (3, 8, 64, 200, (0, 37, 130), 0), (2, 3, 32, 131, (5, 99), 0), (2, 8, 64, 300, (200, 17), 50), (1, 8, 96, 96, (0,), 0)])
def test_rows3_attention_matches_float64_softmax_and_the_fp32_kernel(mods, slots, heads, rps, cache_len, starts, window):
    E, ops = mods
    g = torch.Generator().manual_seed(slots * 100 + rps)
    k = torch.randn(slots, heads, cache_len, 64, generator=g)
    v = torch.randn(slots, heads, cache_len, 64, generator=g)
    q = torch.randn(slots * rps, heads * 64, generator=g) * 2.0
    row_slot = torch.arange(slots).repeat_interleave(rps).int()
    row_pos = torch.cat([torch.arange(rps) + s for s in starts]).int()
    k3, v3 = ops.kv3_encode(k, False).cuda(), ops.kv3_encode(v, True).cuda()
    # positions a row may not see hold large finite junk in the piece caches: they must get zero weight
    got = ops.attention_rows3(q.cuda(), k3, v3, row_pos.cuda(), row_slot.cuda(), rps, heads, cache_len, window).cpu()
    ref = softmax_ref(q, k, v, row_pos, row_slot, heads, window)
    old = ops.attention(q.cuda(), k.cuda(), v.cuda(), row_pos.cuda(), row_slot.cuda(), heads, window).cpu()
    err = ((got.double() - ref).norm() / ref.norm()).item()
    err_old = ((old.double() - ref).norm() / ref.norm()).item()
    print(f"rows3 {err:.2e}, fp32 kernel {err_old:.2e}")
    assert err < 1e-6 and err < 4 * err_old + 2e-7
    assert (got - old).abs().max().item() < 2e-5


@pytest.mark.parametrize("M", [8, 2048])
def test_qkv_epilogue_writes_the_piece_caches(mods, M):
    """EPI_QKV_ROPE with k_cache3 / v_cache3: the pieces hold exactly the fp32 K / V rows the same call wrote (skinny and many-row kernels)."""
    E, ops = mods
    g = torch.Generator().manual_seed(M)
    heads, K, slots, cache_len = 8, 512, 4, M // 4 + 40
    x = torch.randn(M, K, generator=g)
    w = torch.randn(3 * heads * 64, K, generator=g) / math.sqrt(K)
    rope = torch.randn(cache_len, 32, 2, generator=g)
    row_slot = torch.arange(slots).repeat_interleave(M // slots).int()
    row_pos = (torch.arange(M // slots).repeat(slots) + 7).int()
    kc, vc = torch.zeros(slots, heads, cache_len, 64).cuda(), torch.zeros(slots, heads, cache_len, 64).cuda()
    k3, v3 = ops.kv3_cache(slots, heads, cache_len), ops.kv3_cache(slots, heads, cache_len)
    ops.linear(x.cuda(), ops.pack_weight(w, fp32=True), 3 * heads * 64, w_fp32=True, epilogue=E.EPI_QKV_ROPE, rope=rope.cuda(),
               row_pos=row_pos.cuda(), row_slot=row_slot.cuda(), k_cache=kc, v_cache=vc, n_q_heads=heads, n_kv_heads=heads, cache_len=cache_len,
               w3=ops.pack_weight_w3(w), k_cache3=k3, v_cache3=v3)
    assert kc.abs().sum().item() > 0
    assert torch.equal(ops.kv3_decode(k3, cache_len, False), kc.cpu())
    assert torch.equal(ops.kv3_decode(v3, cache_len, True), vc.cpu())
    assert torch.equal(k3.cpu(), ops.kv3_encode(kc.cpu(), False)) and torch.equal(v3.cpu(), ops.kv3_encode(vc.cpu(), True))
