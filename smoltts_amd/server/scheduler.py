"""Continuous batching on top of the slot-restart primitives of the engine.

The reference serves one request at a time (its handlers call the model inline,
mlx_inference/.../server/routes/openai.py:17-28).  Here one worker thread owns an ``LMSession`` with
``max_batch`` slots: new requests are prefilled into free slots while the other slots keep decoding
(``smoltts_lm_prefill`` restarts only the listed slots), every tick decodes a few frames for all slots,
finished slots (``<|im_end|>`` or frame budget) are released.  Prompts longer than ``prefill_chunk`` columns are
prefilled in chunks (``smoltts_lm_prefill_chunk``) with a decode tick for the speaking slots between chunks.  Audio is produced per request by its
own streaming Mimi session, so blocking and streaming responses share one code path and a request's
PCM is identical to what ``SmolTTS.__call__`` / ``stream`` return for it alone.
"""
from __future__ import annotations

import queue
import threading
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class _Request:
    text: str
    voice: str
    stream: bool
    max_new_tokens: int
    out: "queue.Queue" = field(default_factory=queue.Queue)  # np.ndarray chunks, then None (or an Exception)
    slot: int = -1
    emitted: int = 0
    msess: object = None
    pcm: object = None


class BatchScheduler:
    def __init__(self, tts, max_batch: int = 32, frames_per_tick: int = 4, generation_settings=None, max_prompt_rows: int = 4096,
                 prefill_chunk: Optional[int] = 128):
        import torch

        from ..config import GenerationSettings
        from ..engine import LMSession
        from ..generate import _apply_sampling

        self.tts = tts
        self.B = max_batch
        self.tick = frames_per_tick
        self.prefill_chunk = prefill_chunk  # columns per utterance per prefill call (None: whole prompts at once)
        self.settings = generation_settings or GenerationSettings.greedy()
        self.max_frames = self.settings.max_new_tokens + 1
        self.session = LMSession(tts.lm, max_batch, max_seq=tts.config.max_seq_len, max_rows=max(max_prompt_rows, max_batch),
                                 max_frames=self.max_frames)
        _apply_sampling(self.session, self.settings)
        self._torch = torch
        self._pending: "queue.Queue[_Request]" = queue.Queue()
        self._active: Dict[int, _Request] = {}
        self._free: List[int] = list(range(max_batch))
        self._stop = threading.Event()
        self._started = False
        self._thread = threading.Thread(target=self._run, name="smoltts-scheduler", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ client side
    def submit(self, text: str, voice: str = "heart", stream: bool = False, max_new_tokens: Optional[int] = None) -> _Request:
        req = _Request(text, voice, stream, min(max_new_tokens or self.settings.max_new_tokens, self.settings.max_new_tokens))
        self._pending.put(req)
        return req

    def synthesize(self, text: str, voice: str = "heart", max_new_tokens: Optional[int] = None) -> np.ndarray:
        """Blocking: float32 PCM of the whole utterance."""
        return np.concatenate(list(self.iter_chunks(self.submit(text, voice, False, max_new_tokens))) or [np.zeros(0, np.float32)])

    def iter_chunks(self, req: _Request):
        while True:
            item = req.out.get()
            if item is None:
                return
            if isinstance(item, Exception):
                raise item
            yield item

    def close(self) -> None:
        self._stop.set()
        self._thread.join(timeout=30)
        self.session.close()

    # ------------------------------------------------------------------ worker
    def _admit(self) -> None:
        new: List[_Request] = []
        while self._free and not self._pending.empty():
            req = self._pending.get_nowait()
            try:
                req.prompt = self.tts._get_prompt(req.text, req.voice)
                if req.prompt.shape[1] + req.max_new_tokens + 2 > self.session.max_seq:
                    raise ValueError("prompt + max_new_tokens exceed max_seq_len")
            except Exception as e:  # bad request: answer it, keep serving
                req.out.put(e)
                continue
            req.slot = self._free.pop(0)
            new.append(req)
        if not new:
            return
        from ..engine import MimiSession

        if self.prefill_chunk:
            # long prompts (voice-clone speakers) enter in chunks; the slots already speaking get a tick in between
            def between():
                if self._active:
                    self.session.decode(self.tick)
                    self._drain()

            self.session.prefill_chunked([r.prompt for r in new], slots=[r.slot for r in new], stop_on_eos=True,
                                         chunk=self.prefill_chunk, between=between)
        else:
            self.session.prefill([r.prompt for r in new], slots=[r.slot for r in new], stop_on_eos=True)
        self._started = True
        for r in new:
            r.msess = MimiSession(self.tts.codec, max_batch=1, max_chunk_frames=max(self.tick, 1) + 1)
            r.msess.reset()
            self._active[r.slot] = r

    def _drain(self) -> None:
        torch = self._torch
        codes, n_frames, done, _ = self.session.fetch()
        nq = self.tts.config.num_codebooks
        for slot, r in list(self._active.items()):
            n = min(int(n_frames[slot]), r.max_new_tokens + 1)
            # blocking requests keep only frames whose slow id is a semantic token (generate_blocking,
            # lm/generate.py:196-207); streaming requests decode every frame (__init__.py:88-92)
            tc = self.tts.token_config
            sel = [f for f in range(r.emitted, n) if r.stream or tc.semantic_start_id <= codes[slot, f, 0] <= tc.semantic_end_id]
            for i in range(0, len(sel), r.msess.chunk):
                part = sel[i: i + r.msess.chunk]
                k = len(part)
                chunk = torch.from_numpy(np.ascontiguousarray(codes[slot, part][:, -nq:])).reshape(1, k, nq).cuda()
                pcm = torch.empty(1, k * 1920, dtype=torch.float32, device="cuda")
                r.msess.decode_chunk(chunk, 0, k, pcm, code_offset=0)
                r.out.put(pcm.cpu().numpy().reshape(-1))
            r.emitted = n
            if done[slot] or r.emitted >= r.max_new_tokens + 1:
                r.msess.close()
                r.out.put(None)
                del self._active[slot]
                self._free.append(slot)

    def _run(self) -> None:
        try:
            while not self._stop.is_set():
                self._admit()
                if not self._active:
                    try:
                        self._pending.put(self._pending.get(timeout=0.05))  # idle: wait for work without spinning
                    except queue.Empty:
                        pass
                    continue
                self._drain()           # frame 0 of freshly admitted requests / last tick's frames
                if self._active:
                    self.session.decode(self.tick)
        except Exception as e:  # engine failure: fail every waiter loudly
            for r in list(self._active.values()):
                r.out.put(e)
            while not self._pending.empty():
                self._pending.get_nowait().out.put(e)
