"""BASELINE configs[1]: one smoltts_byte_70m stream at B = 1, every frame decoded to PCM and copied to the host as it appears
(the loop bench.py times as `b1_70m_stream_frames_per_s`), taken apart: wall time per frame of (a) the loop as it is, (b) the LM
frame graph alone, (c) the codec step alone, (d) variants: k frames per codec step / per host round trip.
Under `rocprofv3 --kernel-trace` (tools/b1_trace.sh) its kernel table says where the GPU time of a frame goes.
argv: [frames per variant = 128]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from bench import make_prompts  # noqa: E402
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.config import NumericsMode, TokenConfig  # noqa: E402
from smoltts_amd.engine import LMEngine, LMSession, MimiEngine, MimiSession, load_library  # noqa: E402
from smoltts_amd.packing import pack_lm  # noqa: E402
from smoltts_amd.prompt import PromptEncoder  # noqa: E402
from smoltts_amd.synthetic import named_config, synthetic_lm_state  # noqa: E402
from smoltts_amd.tokenizer import load_tokenizer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
load_library()
cfg = named_config("smoltts_byte_70m")
tok = load_tokenizer()
tc = TokenConfig.from_tokenizer(tok, cfg)
num = NumericsMode.torch_reference()
a, o = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), num)
eng = LMEngine(cfg, None, tc, num, arena=a, offsets=o)
meng = MimiEngine(synthetic_mimi_state(seed=3), max_positions=2048)
prompt = make_prompts(PromptEncoder(tok, tc.semantic_start_id), 1)[0]
WARM = 22


def stream(k_codec, k_sync, lm=True, codec=True, fused=True):
    """k_codec frames per codec step, host copy + synchronisation every k_sync frames (multiples of k_codec)."""
    ls = LMSession(eng, max_batch=1, max_seq=600, max_rows=256, max_frames=WARM + N + 8)
    ls.use_fused_depth_attention(fused)
    ms = MimiSession(meng, max_batch=1, max_chunk_frames=k_codec)
    buf = torch.zeros(1, 1920 * (WARM + N + 8), dtype=torch.float32, device="cuda")
    ls.prefill([prompt], stop_on_eos=False)
    ms.reset()
    f, t1 = 1, None
    while f < WARM + N:
        if f >= WARM and t1 is None:
            torch.cuda.synchronize()
            t1, f1 = time.perf_counter(), f
        if lm:
            ls.decode(k_codec)
        if codec:
            ms.decode_chunk(ls.codes, f - 1, k_codec, buf, code_offset=1)
        f += k_codec
        if (f - 1) % k_sync == 0:
            if codec:
                _ = buf[:, (f - 1 - k_sync) * 1920:(f - 1) * 1920].cpu()
            else:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    ms.close(); ls.close()
    return dt / (f - f1) * 1e6


print(f"smoltts_byte_70m, B = 1, {N} timed frames per line; us per frame (wall)")
print(f"  the bench's loop (LM frame + 1-frame codec step + host copy per frame): {stream(1, 1):8.1f}   -> {1e6 / stream(1, 1):6.0f} frames/s")
print(f"  same, depth attention as launches of its own:                          {stream(1, 1, fused=False):8.1f}")
print(f"  LM frames only (host synchronises every frame):                        {stream(1, 1, codec=False):8.1f}")
print(f"  LM frames only, host synchronises every 8 frames:                      {stream(8, 8, codec=False):8.1f}")
print(f"  codec steps only (1 frame per step, copy per frame):                   {stream(1, 1, lm=False):8.1f}")


def facade_loop(overlap):
    from smoltts_amd.generate import stream_pcm

    ls = LMSession(eng, max_batch=1, max_seq=600, max_rows=256, max_frames=WARM + N)
    ms = MimiSession(meng, max_batch=1, max_chunk_frames=1)
    t1 = None
    for f, _ in enumerate(stream_pcm(ls, ms, prompt, stop_on_eos=False, overlap=overlap)):
        if f == WARM - 1:
            t1 = time.perf_counter()
        last = f
    dt = time.perf_counter() - t1
    ms.close(); ls.close()
    return dt / (last - WARM + 1) * 1e6


print(f"  generate.stream_pcm, one stream (what the loop above does):            {facade_loop(False):8.1f}")
t = facade_loop(True)
print(f"  generate.stream_pcm, codec step of frame f beside frame f + 1:         {t:8.1f}   -> {1e6 / t:6.0f} frames/s")
print(f"  2 frames per codec step and host copy:                                 {stream(2, 2):8.1f}")
print(f"  4 frames per codec step and host copy:                                 {stream(4, 4):8.1f}")
print(f"  8 frames per codec step and host copy:                                 {stream(8, 8):8.1f}")
