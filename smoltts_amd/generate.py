"""Generation loop façade over the device-side frame loop.

Mirrors mlx_inference/src/smoltts_mlx/lm/generate.py: ``GenerationSettings`` (:12-16), ``VQToken``
(:19-22), ``SingleBatchGenerator`` (:25-171, an iterator of frames for one utterance) and
``generate_blocking`` (:174-216).  The reference runs one utterance and synchronises with the
device nine times per frame; here the frame loop (slow step, 8 depth steps, argmax, column feedback,
stop rule) runs on the GPU inside ``smoltts_lm_decode`` and the host only fetches finished columns.
``BatchGenerator`` is the same loop for B utterances at once.

Sampling follows the reference's rules (lm/generate.py:88-99,118-132): the slow token is greedy iff
``default_temp == 0``; the depth tokens are sampled iff ``default_fast_temp`` is set and > 0; ``min_p``
(lm/utils/samplers.py:8-34) follows ``GenerationSettings.min_p_mode``: "reference" (default) removes nothing, as the
reference's code does; "intended" keeps tokens with p >= min_p * p_max.  It runs on the device with a counter-based
generator keyed by ``GenerationSettings.seed`` (reproducible for a fixed seed; the reference's MLX
generator is not reproducible across processes, so parity for the sampled modes is statistical).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Any, List, Optional, Sequence

import numpy as np
import torch

from .config import GenerationSettings
from .engine import LMEngine, LMSession


@dataclass
class VQToken:
    semantic_code: int
    audio_codes: Optional[Any]  # (1, n_codebooks, 1) uint32 or None (non-semantic slow id)
    vq_tensor: Any              # (1, 1 + n_fast, 1) uint32


def _apply_sampling(session: LMSession, settings: GenerationSettings) -> None:
    fast = settings.default_fast_temp if settings.default_fast_temp is not None else 0.0
    seed = settings.seed
    if seed is None:
        import os

        seed = int.from_bytes(os.urandom(8), "little")
    session.set_sampling(temp=settings.default_temp, fast_temp=max(fast, 0.0), min_p=settings.effective_min_p, seed=seed)


def _frame_to_token(engine: LMEngine, col: np.ndarray) -> VQToken:
    """lm/generate.py:143-159: audio codes exist only when the slow id is a semantic token."""
    tc, cfg = engine.token_config, engine.cfg
    slow = int(col[0])
    vq = col.astype(np.uint32)[None, :, None]
    audio = None
    if tc.semantic_end_id is not None and tc.semantic_start_id <= slow <= tc.semantic_end_id:
        codes = col[1:] if cfg.duplicate_code_0 else np.concatenate([[slow - tc.semantic_start_id], col[1:]])
        audio = codes.astype(np.uint32)[None, :, None]
    return VQToken(semantic_code=slow, audio_codes=audio, vq_tensor=vq)


class BatchGenerator:
    """Frames for B utterances; ``next()`` returns a list with one VQToken (or None once the
    utterance has stopped) per slot.  ``frames_per_sync`` > 1 lets the GPU run ahead."""

    def __init__(self, engine: LMEngine, prompts: Sequence[np.ndarray], generation_settings: GenerationSettings,
                 audio_only: bool = True, frames_per_sync: int = 1, session: Optional[LMSession] = None):
        self.engine = engine
        self.settings = generation_settings
        self.audio_only = audio_only
        self.B = len(prompts)
        max_new = generation_settings.max_new_tokens if generation_settings.max_new_tokens is not None else engine.cfg.max_seq_len
        self.max_frames = max_new + 1  # input_pos counts 0..max_new_tokens inclusive (generate.py:60,161)
        max_T = max(int(p.shape[1]) for p in prompts)
        self.session = session or LMSession(engine, self.B, max_seq=min(engine.cfg.max_seq_len, max_T + self.max_frames + 1),
                                            max_rows=sum(int(p.shape[1]) for p in prompts), max_frames=self.max_frames)
        _apply_sampling(self.session, generation_settings)
        self._prompts = list(prompts)
        self._started = False
        self._emitted = np.zeros(self.B, dtype=np.int64)
        self._per_sync = max(1, frames_per_sync)
        self._pending: List[List[Optional[VQToken]]] = []
        # wall-clock figures of the reference (prefill ms, frames/s, x realtime: lm/generate.py:187-214), from events on the
        # launch stream so that measuring does not add a synchronisation
        self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]  # before prefill, after it, after the last decode

    def __iter__(self):
        return self

    def _run(self) -> None:
        s = self.session
        if not self._started:
            self._ev[0].record()
            s.prefill(self._prompts, stop_on_eos=self.audio_only)
            self._ev[1].record()
            self._started = True
            if self._per_sync > 1:
                s.decode(self._per_sync - 1)
        else:
            s.decode(self._per_sync)
        self._ev[2].record()
        codes, n_frames, done, _ = s.fetch()
        top = int(n_frames.max())
        lo = int(self._emitted.min()) if (n_frames > self._emitted).any() else top
        for f in range(lo, top):
            row: List[Optional[VQToken]] = []
            any_new = False
            for b in range(self.B):
                if self._emitted[b] <= f < n_frames[b]:
                    row.append(_frame_to_token(self.engine, codes[b, f]))
                    any_new = True
                else:
                    row.append(None)
            if any_new:
                self._pending.append(row)
        self._emitted = np.maximum(self._emitted, n_frames)
        self._all_done = bool(done.all())

    def __next__(self) -> List[Optional[VQToken]]:
        if not self._pending:
            if self._started and getattr(self, "_all_done", False):
                raise StopIteration
            self._run()
            if not self._pending:
                raise StopIteration
        return self._pending.pop(0)

    def stats(self) -> dict:
        """Prefill time (prompt -> frame 0) and decode rate of what has run so far; the rate is the reference's
        "x realtime": frames after the first / 12.5 / decode seconds, summed over the batch (prefill and codec excluded)."""
        if not self._started:
            return {}
        torch.cuda.current_stream().synchronize()
        prefill_ms = self._ev[0].elapsed_time(self._ev[1])
        decode_s = self._ev[1].elapsed_time(self._ev[2]) / 1e3
        frames = int(self._emitted.sum())
        after_first = max(frames - self.B, 0)
        rate = after_first / decode_s if decode_s > 0 and after_first else 0.0
        n_prompt = sum(int(p.shape[1]) for p in self._prompts)
        return {"utterances": self.B, "prompt_tokens": n_prompt, "prefill_ms": prefill_ms,
                "prefill_tokens_per_s": n_prompt / (prefill_ms / 1e3) if prefill_ms > 0 else 0.0, "frames": frames,
                "decode_s": decode_s, "frames_per_s": rate, "ms_per_frame_step": 1e3 * decode_s / max(after_first / self.B, 1e-9) if after_first else 0.0,
                "realtime_x": rate / 12.5}

    def close(self) -> None:
        self.session.close()


class SingleBatchGenerator:
    """One utterance, one frame per ``next()`` (reference class of the same name)."""

    def __init__(self, model: LMEngine, prompt: np.ndarray, generation_settings: GenerationSettings, audio_only: bool = True):
        prompt = np.asarray(prompt)
        if prompt.ndim == 3:
            prompt = prompt[0]
        self._inner = BatchGenerator(model, [prompt], generation_settings, audio_only=audio_only)

    def __iter__(self):
        return self

    def __next__(self) -> VQToken:
        return next(self._inner)[0]

    def close(self) -> None:
        self._inner.close()


def generate_blocking(model: LMEngine, prompt: np.ndarray, generation_settings: GenerationSettings, audio_only: bool = True) -> np.ndarray:
    """-> (1, n_codebooks, F) uint32 audio codes (frames whose slow id is not semantic are dropped,
    lm/generate.py:196-207), or the (1, 1+n_fast, F) columns when ``audio_only`` is False."""
    prompt = np.asarray(prompt)
    if prompt.ndim == 3:
        prompt = prompt[0]
    gen = BatchGenerator(model, [prompt], generation_settings, audio_only=audio_only, frames_per_sync=16)
    out = []
    for row in gen:
        tok = row[0]
        if tok is None:
            continue
        if audio_only:
            if tok.audio_codes is not None:
                out.append(tok.audio_codes)
        else:
            out.append(tok.vq_tensor)
    gen.close()
    if not out:
        return np.zeros((1, model.cfg.num_codebooks if audio_only else model.grid_height, 0), dtype=np.uint32)
    return np.concatenate(out, axis=-1)


def stream_pcm(session: LMSession, msession, prompt: np.ndarray, stop_on_eos: bool = True, max_frames: Optional[int] = None,
               overlap: bool = True) -> Iterator[np.ndarray]:
    """One utterance in slot 0 of ``session`` -> one 1920-sample float32 chunk per generated frame, as the reference's
    ``SmolTTS.stream`` yields them (mlx_inference/src/smoltts_mlx/__init__.py:83-95: every frame of ``SingleBatchGenerator`` through
    ``codec.decode_step``), the terminating ``<|im_end|>`` frame included.

    At one utterance both halves of a frame are chains of dependent launches that leave the chip almost empty (188 LM launches,
    ~100 codec launches: DESIGN.md 5), so with ``overlap`` they run side by side: frame f + 1 is queued on the LM stream before
    the codec step of frame f goes out on a second stream -- the host has waited for frame f's event by then, it never parks a
    device-side wait in a queue.  Every chunk still leaves as soon as its own codec step has run; the numbers are those of the
    one-stream loop (``overlap=False``), only the order in which the GPU sees the launches changes.  A frame queued behind the
    last one is not a frame: a stopped slot is frozen (smoltts_lm_decode)."""
    s = session
    limit = s.max_frames if max_frames is None else min(max_frames, s.max_frames)
    dev = s.engine.device
    lm_stream = torch.cuda.Stream(dev) if overlap else torch.cuda.current_stream(dev)
    codec_stream = torch.cuda.Stream(dev) if overlap else lm_stream
    caller = torch.cuda.current_stream(dev)
    lm_stream.wait_stream(caller)
    codec_stream.wait_stream(caller)
    pcm_dev = torch.empty(1, 1920, dtype=torch.float32, device=dev)
    pcm_host = torch.empty(1, 1920, dtype=torch.float32).pin_memory()
    state_host = torch.zeros(2, dtype=torch.int32).pin_memory()  # n_frames[0], done[0]
    with torch.cuda.stream(codec_stream):
        msession.reset()
    with torch.cuda.stream(lm_stream):
        s.prefill([prompt], stop_on_eos=stop_on_eos)  # frame 0
        ev = torch.cuda.Event()
        ev.record(lm_stream)
    f = 0
    try:
        while f < limit:
            nxt = None
            if overlap and f + 1 < limit:
                with torch.cuda.stream(lm_stream):
                    s.decode(1)  # frame f + 1 runs beside the codec step of frame f
                    nxt = torch.cuda.Event()
                    nxt.record(lm_stream)
            ev.synchronize()  # frame f is in the output ring
            with torch.cuda.stream(codec_stream):
                state_host[0:1].copy_(s.n_frames[0:1], non_blocking=True)
                state_host[1:2].copy_(s.done[0:1], non_blocking=True)
                msession.decode_chunk(s.codes[:, f:f + 1], 0, 1, pcm_dev, code_offset=1)  # (decode_chunk writes frame f0 at pcm[:, 1920 f0:])
                pcm_host.copy_(pcm_dev, non_blocking=True)
            codec_stream.synchronize()
            n, done = int(state_host[0]), int(state_host[1])
            if n <= f:  # the slot had stopped before this frame
                break
            yield pcm_host.numpy().reshape(-1).copy()
            if done and n == f + 1:
                break
            if not overlap and f + 1 < limit:
                s.decode(1)
                nxt = torch.cuda.Event()
                nxt.record(lm_stream)
            if nxt is None:
                break
            ev = nxt
            f += 1
    finally:
        lm_stream.synchronize()
        codec_stream.synchronize()
