"""Decode attention with the keys of a (row, kv head) pair on two workgroups, merged inside the launch (attn_split_kernel),
and the 8-lanes-per-key read of a bf16 cache -- against softmax(q K^T / 8) V in fp32 on the CPU
(modeling/model/rq_transformer.py:554-568; cache semantics lm/cache.py:6-22).

The split kernel keeps state across launches (arrival tickets count up, the partial records are reused): the same scratch goes
through many launches with changing rows, including rows that have nothing cached, and must stay exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, kc, vc, row_pos, row_slot, Hq, window):
    Hkv = kc.shape[1]
    G = Hq // Hkv
    out = torch.zeros_like(q)
    for r in range(q.shape[0]):
        p, s = int(row_pos[r]), int(row_slot[r])
        if p < 0 or p >= kc.shape[2]:
            continue
        lo = max(0, p + 1 - window) if window else 0
        K = kc[s, :, lo: p + 1].float().repeat_interleave(G, dim=0)
        V = vc[s, :, lo: p + 1].float().repeat_interleave(G, dim=0)
        out[r] = (torch.softmax(q[r].view(Hq, 1, 64) @ K.transpose(1, 2) / 8.0, dim=-1) @ V).reshape(-1)
    return out


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))


@pytest.mark.parametrize("Hq,Hkv", [(12, 4), (9, 3), (8, 8), (8, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_split_attention_matches_fp32_softmax(Hq, Hkv, dtype):
    from smoltts_amd import ops

    g = torch.Generator().manual_seed(Hq * 7 + (dtype == torch.bfloat16))
    slots, cache_len = 32, 1100
    kc = torch.randn(slots, Hkv, cache_len, 64, generator=g).to(dtype)
    vc = torch.randn(slots, Hkv, cache_len, 64, generator=g).to(dtype)
    part = torch.full((ops.SPLIT_PART_FLOATS,), float("nan"), device="cuda")  # poisoned: only this launch's records may be read
    ticket = torch.zeros(ops.SPLIT_TICKETS, dtype=torch.int32, device="cuda")
    kcd, vcd = kc.cuda(), vc.cuda()
    cases = [
        torch.tensor([0, 1, 3, 17, 63, 64, 127, 128, 129, 255, 256, 511, 512, 600, 1023, 1099], dtype=torch.int32),  # every boundary of the key deal
        torch.randint(0, cache_len, (128 // Hkv,), generator=g, dtype=torch.int32),                                 # the full 128 pairs
        torch.tensor([5, -1, cache_len, 900, 2, cache_len + 7, 330], dtype=torch.int32),                            # rows with nothing cached
        torch.randint(200, 1000, (7,), generator=g, dtype=torch.int32),
        torch.tensor([510, 511, 512, 513, 760, 761], dtype=torch.int32),                                             # around the split threshold (also with window 250: below it)
    ]
    for rep in range(3):  # tickets and records live on across launches
        for row_pos in cases:
            rows = row_pos.numel()
            row_slot = torch.randint(0, slots, (rows,), generator=g, dtype=torch.int32)  # (several rows may read one slot's cache)
            q = torch.randn(rows, Hq * 64, generator=g)
            for window in (0, 250):
                out = ops.attention_split(q.cuda(), kcd, vcd, row_pos.cuda(), row_slot.cuda(), Hq, (part, ticket), window).cpu()
                ref = _ref(q, kc, vc, row_pos, row_slot, Hq, window)
                assert _rel(out, ref) < 2e-5, (Hq, Hkv, dtype, rep, window, row_pos.tolist())
    assert int(ticket.max()) > 0 and int((ticket % 2).sum()) == 0  # every pair that took tickets took them in twos (rows below 512 keys take none)


@pytest.mark.parametrize("Hq,Hkv", [(12, 4), (9, 3)])
def test_bf16_cache_without_split_uses_the_eight_lane_kernel(Hq, Hkv):
    """A bf16 cache always runs attn_split_kernel (8 lanes per key); with more than 128 pairs it runs unsplit."""
    from smoltts_amd import ops

    g = torch.Generator().manual_seed(3)
    slots, cache_len, rows = 64, 700, 64  # 64 x Hkv pairs > 128
    kc = torch.randn(slots, Hkv, cache_len, 64, generator=g).to(torch.bfloat16)
    vc = torch.randn(slots, Hkv, cache_len, 64, generator=g).to(torch.bfloat16)
    row_pos = torch.randint(0, cache_len, (rows,), generator=g, dtype=torch.int32)
    row_slot = torch.arange(rows, dtype=torch.int32)
    q = torch.randn(rows, Hq * 64, generator=g)
    out = ops.attention(q.cuda(), kc.cuda(), vc.cuda(), row_pos.cuda(), row_slot.cuda(), Hq).cpu()
    assert _rel(out, _ref(q, kc, vc, row_pos, row_slot, Hq, 0)) < 2e-5


def test_session_ids_do_not_depend_on_the_split():
    """A 70m session decoded with and without the key split emits the same ids (its smallest top-2 gap is far above fp32 noise)."""
    import numpy as np

    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    cfg = named_config("smoltts_byte_70m")
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    eng = LMEngine(cfg, synthetic_lm_state(cfg, seed=4), tc)
    pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
    prompts = [pe.build_prompt("split or not, " * (3 + 5 * b), "heart") for b in range(4)]  # contexts on both sides of 128
    grids = []
    for on in (True, False):
        s = LMSession(eng, max_batch=4, max_seq=512, max_rows=1024, max_frames=40)
        s.use_split_attention(on)
        s.prefill(prompts, stop_on_eos=False)
        s.decode(39)
        codes, n, _, margin = s.fetch()
        grids.append((codes[:, :40].copy(), margin.copy()))
        s.close()
    assert float(grids[1][1].min()) > 1e-6
    assert np.array_equal(grids[0][0], grids[1][0])
    eng.close()
