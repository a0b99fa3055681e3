"""The HIP engine decoding from the checkpoint directory the reference's ``save_pretrained`` wrote (tests/golden/ref_ckpt_micro/,
see tests/test_ref_checkpoint_cpu.py): ids bit-identical to ``lm_ref_ckpt_micro.npz``, whose grids are pinned on the reference
model read back by its own ``from_pretrained``.  A one-head, 64-wide, 64-entry-codebook model also exercises the kernels' smallest shapes."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("fused", [True, False])
def test_engine_decodes_the_reference_written_checkpoint(fused):
    from smoltts_amd.checkpoint import load_checkpoint
    from smoltts_amd.config import TokenConfig
    from smoltts_amd.engine import LMEngine, LMSession

    cfg, tok, state = load_checkpoint(GOLD / "ref_ckpt_micro")
    g = np.load(GOLD / "lm_ref_ckpt_micro.npz")
    tc = TokenConfig.from_tokenizer(tok, cfg)
    eng = LMEngine(cfg, state, tc)
    F = int(g["frames"])
    prompts = [g["prompt_0"], g["prompt_1"]]
    s = LMSession(eng, max_batch=2, max_seq=256, max_rows=256, max_frames=F)
    s.use_fused_depth_attention(fused)
    s.prefill(prompts, stop_on_eos=False)
    s.decode(F - 1)
    codes, n, _, margin = s.fetch()
    for b in range(2):
        assert n[b] == F and np.array_equal(codes[b, :F].T, g[f"grid_{b}"]), f"slot {b}: ids differ from the reference-pinned grid"
    s.close()
    eng.close()

