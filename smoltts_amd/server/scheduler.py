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
    pending: list = field(default_factory=list)  # blocking requests: audio-code columns awaiting one batched codec call


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
        self._codec_pool: Dict[int, list] = {}  # chunk size -> idle one-slot codec sessions (their slabs are re-used)
        self._finished: List[_Request] = []     # complete blocking utterances waiting for their (batched) codec pass
        self._finished_age = 0
        self._batch_codec = None                # one multi-slot codec session for those passes
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
        for pool in self._codec_pool.values():
            for sess in pool:
                sess.close()
        self._codec_pool.clear()
        if self._batch_codec is not None:
            self._batch_codec.close()
            self._batch_codec = None
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
            # streaming answers decode every tick through a one-slot codec session; blocking answers are decoded when
            # the utterance is complete, several utterances per codec pass (_decode_finished)
            r.msess = self._codec_session(max(self.tick, 1) + 1) if r.stream else None
            self._active[r.slot] = r

    def _codec_session(self, chunk: int):
        from ..engine import MimiSession

        pool = self._codec_pool.setdefault(chunk, [])
        sess = pool.pop() if pool else MimiSession(self.tts.codec, max_batch=1, max_chunk_frames=chunk)
        sess.reset()
        return sess

    def _release_codec(self, sess) -> None:
        self._codec_pool.setdefault(sess.chunk, []).append(sess)

    def _decode_frames(self, r: _Request, cols: np.ndarray) -> None:
        """cols (k, nq) audio codes of consecutive frames of request r -> PCM chunks on its queue."""
        torch = self._torch
        nq = cols.shape[1]
        for i in range(0, cols.shape[0], r.msess.chunk):
            part = np.ascontiguousarray(cols[i: i + r.msess.chunk])
            k = part.shape[0]
            chunk = torch.from_numpy(part).reshape(1, k, nq).cuda()
            pcm = torch.empty(1, k * 1920, dtype=torch.float32, device="cuda")
            r.msess.decode_chunk(chunk, 0, k, pcm, code_offset=0)
            r.out.put(pcm.cpu().numpy().reshape(-1))

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
            r.emitted = n
            finished = bool(done[slot]) or r.emitted >= r.max_new_tokens + 1
            if sel:
                cols = codes[slot, sel][:, -nq:].astype(np.int32)
                if r.stream:
                    self._decode_frames(r, cols)
                else:
                    r.pending.append(cols)
            if finished:
                del self._active[slot]
                self._free.append(slot)
                if r.stream:
                    self._release_codec(r.msess)
                    r.msess = None
                    r.out.put(None)
                else:
                    self._finished.append(r)

    CODEC_BATCH, CODEC_CHUNK = 8, 64

    def _decode_finished(self, force: bool) -> None:
        """Codec pass over up to CODEC_BATCH complete blocking utterances at once (each in its own slot from position 0,
        the shorter ones padded at the end: the decoder is causal, so the padding cannot reach their samples).  Runs when
        enough utterances wait, when they have waited two ticks, or when nothing else is going on."""
        if not self._finished:
            return
        self._finished_age += 1
        if not (force or len(self._finished) >= self.CODEC_BATCH or self._finished_age >= 2):
            return
        torch = self._torch
        from ..engine import MimiSession

        if self._batch_codec is None:
            self._batch_codec = MimiSession(self.tts.codec, max_batch=self.CODEC_BATCH, max_chunk_frames=self.CODEC_CHUNK)
        nq = self.tts.config.num_codebooks
        while self._finished:
            batch, self._finished = self._finished[: self.CODEC_BATCH], self._finished[self.CODEC_BATCH:]
            cols = [np.concatenate(r.pending) if r.pending else np.zeros((0, nq), np.int32) for r in batch]
            F = max(c.shape[0] for c in cols)
            if F > 0:
                grid = np.zeros((len(batch), F, nq), np.int32)
                for b, c in enumerate(cols):
                    grid[b, : c.shape[0]] = c
                pcm = self._batch_codec.decode(torch.from_numpy(grid).cuda()).cpu().numpy()
                for b, (r, c) in enumerate(zip(batch, cols)):
                    if c.shape[0]:
                        r.out.put(pcm[b, : c.shape[0] * 1920].copy())
            for r in batch:
                r.pending = []
                r.out.put(None)
        self._finished_age = 0

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
                    self.session.decode(self.tick)  # asynchronous: the codec pass below queues behind it
                self._decode_finished(force=not self._active)
        except Exception as e:  # engine failure: fail every waiter loudly
            for r in list(self._active.values()) + self._finished:
                r.out.put(e)
            while not self._pending.empty():
                self._pending.get_nowait().out.put(e)
