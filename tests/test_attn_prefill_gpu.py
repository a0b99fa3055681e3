"""Many-row attention on the matrix cores (prompt prefill, codec transformer): packed utterances with boundaries inside
groups of four rows, arbitrary row->slot orders, windows, rows without cache, X3 output."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from smoltts_amd import engine, ops

    engine.load_library()
    return ops


def rel_err(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def ref_attention(q, kc, vc, pos, slot, Hq, window):
    Hkv = kc.shape[1]
    G = Hq // Hkv
    out = torch.zeros(q.shape[0], Hq * 64, dtype=torch.float64)
    for r in range(q.shape[0]):
        p, s = int(pos[r]), int(slot[r])
        if p < 0 or p >= kc.shape[2]:
            continue
        lo = max(0, p + 1 - window) if window else 0
        K = kc[s, :, lo: p + 1].double().repeat_interleave(G, dim=0)
        V = vc[s, :, lo: p + 1].double().repeat_interleave(G, dim=0)
        a = torch.softmax(q[r].view(Hq, 1, 64).double() @ K.transpose(1, 2) / 8.0, dim=-1) @ V
        out[r] = a.reshape(-1)
    return out


# with and without GQA, windows, utterance boundaries at every residue
@pytest.mark.parametrize("Hq,Hkv,window", [(8, 8, 0), (8, 8, 50), (4, 4, 7), (12, 4, 0), (9, 3, 0), (6, 3, 20), (8, 2, 0), (12, 4, 5)])
def test_packed_prompts(ops, Hq, Hkv, window):
    g = torch.Generator().manual_seed(Hq + window)
    lengths = [1, 2, 3, 5, 67, 130, 4, 33, 64, 1, 95]  # utterance boundaries at every residue mod 4
    lengths += [40] * max(0, (1100 // Hkv - sum(lengths)) // 40 + 1)  # enough rows for the many-row path
    slots, cache_len = len(lengths), 140
    kc, vc = torch.randn(slots, Hkv, cache_len, 64, generator=g), torch.randn(slots, Hkv, cache_len, 64, generator=g)
    order = torch.randperm(slots, generator=g).tolist()  # packed in an arbitrary slot order
    pos = torch.cat([torch.arange(lengths[s]) for s in order]).int()
    slot = torch.cat([torch.full((lengths[s],), s) for s in order]).int()
    rows = pos.numel()
    assert rows * Hkv >= 1024
    pos[7] = -1           # a row with nothing cached
    pos[100] = cache_len  # out of range -> zero output, no fault
    q = torch.randn(rows, Hq * 64, generator=g)
    x3 = ops.x3_alloc(rows, Hq * 64)
    out = ops.attention(q.cuda(), kc.cuda(), vc.cuda(), pos.cuda(), slot.cuda(), Hq, window, out_x3=x3).cpu()
    ref = ref_attention(q, kc, vc, pos, slot, Hq, window)
    assert rel_err(out.double(), ref) < 1e-5
    assert torch.equal(ops.x3_to_float(x3, rows, Hq * 64), out)
    assert float(out[7].abs().max()) == 0.0 and float(out[100].abs().max()) == 0.0


def test_codec_like_rows_with_offsets(ops):
    """64 new positions per slot appended behind 200 cached ones (the streaming codec transformer's shape)."""
    g = torch.Generator().manual_seed(3)
    Hq = Hkv = 8
    slots, cache_len, new, old = 20, 300, 64, 200
    kc, vc = torch.randn(slots, Hkv, cache_len, 64, generator=g), torch.randn(slots, Hkv, cache_len, 64, generator=g)
    pos = (old + torch.arange(new)).repeat(slots).int()
    slot = torch.arange(slots).repeat_interleave(new).int()
    q = torch.randn(slots * new, Hq * 64, generator=g)
    out = ops.attention(q.cuda(), kc.cuda(), vc.cuda(), pos.cuda(), slot.cuda(), Hq, 0).cpu()
    assert rel_err(out.double(), ref_attention(q, kc, vc, pos, slot, Hq, 0)) < 1e-5
