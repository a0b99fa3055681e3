"""Continuous batching on top of the slot-restart primitives of the engine.

The reference serves one request at a time (its handlers call the model inline,
mlx_inference/.../server/routes/openai.py:17-28).  Here one worker thread owns an ``LMSession`` with
``max_batch`` slots: new requests are prefilled into free slots while the other slots keep decoding
(``smoltts_lm_prefill_deferred`` restarts only the listed slots and leaves frame 0 to the next frame graph), every tick
decodes a few frames for all slots, finished slots (``<|im_end|>`` or frame budget) are released.  Prompts longer than
``prefill_chunk`` columns are prefilled in chunks (``smoltts_lm_prefill_chunk``) with a tick for the speaking slots
between chunks.

The loop is pipelined so that the GPU never waits for Python: tick k is queued, a device-side snapshot of the output ring
is queued behind it, and *then* the host reads the snapshot of tick k-1 (on a copy stream) and does its bookkeeping while
tick k runs.  A slot that finished in tick k-1 is therefore refilled in tick k+1 (one tick of that slot is the price).
Codec work is queued on the same stream behind the ticks and its PCM is fetched on the copy stream when its event has
fired: streaming requests share one codec session whose slots mirror the LM slots (every slot keeps its own stream
position, ``smoltts_mimi_reset_slots`` starts a new stream in a slot) and are decoded together, one codec pass per tick; a
blocking request is decoded when its utterance is complete, up to ``CODEC_BATCH`` finished utterances per codec pass.  A request's PCM is identical to what
``SmolTTS.__call__`` / ``stream`` return for it alone.
"""
from __future__ import annotations

import queue
import threading
import time
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
    pending: list = field(default_factory=list)  # blocking requests: audio-code columns awaiting their codec pass
    prompt: object = None
    first_tick: int = 0  # index of the tick that produces this request's frame 0: older snapshots show the slot's previous tenant
    last_tick: int = 1 << 62  # index of the tick in which the frame budget runs out (known at admission): the slot's next tenant may
                              # be queued right behind it, before the host has seen that tick's snapshot
    retired: bool = False     # the slot has been handed to the next tenant; later snapshots of the slot are not this request's
    stream_done: bool = False  # streaming: the last frames have been seen (the end marker goes out with the last codec pass)
    cancelled: bool = False  # set by the client side (cancel / an abandoned chunk iterator), honoured by the worker at its next look
    closed: bool = False  # the end marker (None or an exception) has been queued


@dataclass
class _CodecJob:
    req: _Request
    cols: np.ndarray  # (F, n_codebooks) int32 audio codes of the complete utterance
    done: int = 0     # frames already handed to the codec


class BatchScheduler:
    CODEC_BATCH, CODEC_CHUNK, CODEC_WAIT = 16, 64, 32  # slots, frames per slot and pass, LM frames a pass may wait for company

    def __init__(self, tts, max_batch: int = 32, frames_per_tick: int = 4, generation_settings=None, max_prompt_rows: int = 4096,
                 prefill_chunk: Optional[int] = 128, overlap_stream_codec: bool = True, side_prefill: bool = True,
                 side_prefill_min_active: Optional[int] = None, codec_products: int = 6):
        import torch

        from ..config import GenerationSettings
        from ..engine import LMSession
        from ..generate import _apply_sampling

        self.tts = tts
        self.B = max_batch
        self.codec_products = codec_products  # SMOLTTS_MIMI_OPT_PRODUCTS of the codec sessions (6: fp32-grade)
        self.tick = frames_per_tick
        self.prefill_chunk = prefill_chunk  # columns per utterance per prefill call (None: whole prompts at once)
        # streaming requests: the codec pass of tick k is launched on a second stream by the host once it has seen tick k finish
        # (it waits for that anyway, to read the tick's snapshot), i.e. while tick k+1 runs -- instead of in line between the ticks
        self.overlap_stream_codec = overlap_stream_codec
        # refills while most slots are speaking: the new prompts' KV rows are computed on a second stream beside the next tick
        # (LMSession.side_park / side_run / side_start) instead of in line between two ticks; the new tenants then start one
        # tick later.  With few slots speaking the in-line prefill answers sooner and stops nobody worth mentioning.
        self.side_prefill = side_prefill and __import__("os").environ.get("SMOLTTS_SIDE_PREFILL") != "0"  # (the switch of tools/bench_scheduler.py A/B runs)
        self.side_min_active = max(1, max_batch // 2) if side_prefill_min_active is None else side_prefill_min_active
        self._side = None  # the one refill in flight: {"h": handle, "reqs": [...], "state": "parked" | "running"}
        self.settings = generation_settings or GenerationSettings.greedy()
        self.max_frames = self.settings.max_new_tokens + 1
        self.session = LMSession(tts.lm, max_batch, max_seq=tts.config.max_seq_len, max_rows=max(max_prompt_rows, max_batch),
                                 max_frames=self.max_frames)
        _apply_sampling(self.session, self.settings)
        self.session.set_frames_per_graph(min(max(frames_per_tick, 1), 8))  # a tick is one graph launch where it fits
        self._torch = torch
        self._pending: "queue.Queue[_Request]" = queue.Queue()
        self._active: Dict[int, _Request] = {}
        self._retiring: List[_Request] = []     # budget-terminated requests whose slot already has its next tenant; their last
                                                # snapshots are still to be read
        self._free: List[int] = list(range(max_batch))
        self._stop = threading.Event()
        self._wake = threading.Event()          # set by submit(): the idle worker sleeps on it (no peeking into the queue)
        self._finished: List[_Request] = []     # complete blocking utterances waiting for a codec slot
        self._batch_codec = None                # codec session of the blocking requests: CODEC_BATCH slots, one utterance each
        self._codec_jobs: List[Optional[_CodecJob]] = [None] * self.CODEC_BATCH
        self._codec_fresh: List[int] = []       # slots whose utterance has not been through a pass yet (stream restart due)
        self._codec_slot_age = [0] * self.CODEC_BATCH  # passes a slot has seen since its last restart
        self._codec_wait = 0                    # ticks since the last pass while work was waiting
        self._stream_codec = None               # codec session whose slot b carries the stream of LM slot b (streaming requests)
        self._codec_age = [0] * max_batch       # codec passes each of its slots has seen since that slot's last reset
        self._deliveries: List[tuple] = []      # (event, pcm on the device, [(request, first sample, n samples, last?)]) in order
        self._snaps: List[tuple] = []           # snapshots of the output ring the host has not looked at yet (oldest first)
        self._tick_no = 0                       # ticks queued so far
        self._counts = {"completed": 0, "cancelled": 0, "failed": 0, "frames_delivered": 0}
        self._dead: Optional[Exception] = None  # why the worker stopped
        self._draining = False
        self._held: Optional[_Request] = None   # next in line, waiting for room in a prefill call
        self._gpu_wait_s = 0.0                  # time the worker spent waiting for the GPU (small => the host is the limit)
        self._t0 = time.time()
        self._thread = threading.Thread(target=self._run, name="smoltts-scheduler", daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ client side
    def submit(self, text: str, voice: str = "heart", stream: bool = False, max_new_tokens: Optional[int] = None) -> _Request:
        if self._dead is not None:  # the worker is gone (engine failure or close): nobody would ever answer
            raise RuntimeError(f"scheduler is not running: {self._dead}")
        if self._draining:
            raise RuntimeError("scheduler is not running: shutting down")
        req = _Request(text, voice, stream, min(max_new_tokens or self.settings.max_new_tokens, self.settings.max_new_tokens))
        self._pending.put(req)
        self._wake.set()
        if self._dead is not None:  # lost the race with a failing worker: answer it ourselves
            self._end(req, RuntimeError(f"scheduler is not running: {self._dead}"))
        return req

    def synthesize(self, text: str, voice: str = "heart", max_new_tokens: Optional[int] = None) -> np.ndarray:
        """Blocking: float32 PCM of the whole utterance."""
        return np.concatenate(list(self.iter_chunks(self.submit(text, voice, False, max_new_tokens))) or [np.zeros(0, np.float32)])

    def iter_chunks(self, req: _Request):
        """Chunks of one request; abandoning the iterator (a client that went away mid-stream) cancels the request."""
        ended = False
        try:
            while True:
                item = req.out.get()
                if item is None or isinstance(item, Exception):
                    ended = True
                    if item is None:
                        return
                    raise item
                yield item
        finally:
            if not ended:
                self.cancel(req)

    def cancel(self, req: _Request) -> None:
        """Stop working for ``req``: its slot is handed to the next request at the worker's next look (within a tick); a
        request that is still queued never starts.  The slot itself simply keeps decoding until it is restarted — a batch
        step costs the same with or without it.  Nothing more is delivered except the end marker."""
        req.cancelled = True

    def close(self, drain: bool = False, timeout: float = 300.0) -> None:
        """Stop the worker.  ``drain``: take no new requests but finish the ones in the books first (a server shutting down);
        otherwise requests in flight are answered with an error."""
        if drain and self._dead is None:
            self._draining = True
            deadline = time.time() + timeout
            while (self._thread.is_alive() and time.time() < deadline and
                   (self._active or self._side is not None or self._retiring or self._held is not None or not self._pending.empty() or self._codec_backlog() or self._deliveries)):
                time.sleep(0.01)
        self._stop.set()
        self._wake.set()
        self._thread.join(timeout=30)
        for name in ("_batch_codec", "_stream_codec"):
            if getattr(self, name) is not None:
                getattr(self, name).close()
                setattr(self, name, None)
        self.session.close()

    # ------------------------------------------------------------------ worker: admission
    def _side_advance(self) -> None:
        """The refill in flight, one step on: parked -> its side call goes out (the host waits for the park, a few us behind the
        tick it was queued after); running -> the slots are armed on the frame stream and the requests enter the books: their
        frame 0 comes out of the next tick."""
        sd = self._side
        if sd is None:
            return
        if sd["state"] == "parked":
            with self._torch.cuda.stream(self._side_stream):
                t = time.perf_counter()
                self.session.side_run(sd["h"])
                self._gpu_wait_s += time.perf_counter() - t
            sd["state"] = "running"
            return
        t = time.perf_counter()
        self.session.side_start(sd["h"], stop_on_eos=True)
        self._gpu_wait_s += time.perf_counter() - t
        self._side = None
        self._enter(sd["reqs"])

    def _enter(self, new: List[_Request]) -> None:
        """Requests whose slots have just been armed: codec stream bookkeeping, first / last tick, into the active set."""
        streams = [r.slot for r in new if r.stream]
        if streams:
            from ..engine import MimiSession

            if self._stream_codec is None:
                self._stream_codec = MimiSession(self.tts.codec, max_batch=self.B, max_chunk_frames=max(self.tick, 1), products=self.codec_products)
                self._stream_codec.reset()
            if not self.overlap_stream_codec:  # (overlapped: the slot's stream restarts on the codec stream, right before the pass
                self._stream_codec.reset_slots(streams)  # of the request's first tick and behind the previous tenant's last pass)
                for b in streams:
                    self._codec_age[b] = 0
        for r in new:
            r.first_tick = self._tick_no
            r.last_tick = self._tick_no + -(-(r.max_new_tokens + 1) // self.tick) - 1  # ceil(frames / tick) ticks from first_tick on
            self._active[r.slot] = r

    def _admit(self) -> None:
        if self._side is not None and self._side["state"] == "running":
            self._side_advance()
        new: List[_Request] = []
        rows = 0  # prompt rows of the (first) prefill call of this admission; the session's workspace holds max_rows
        side = self.side_prefill and self._side is None and len(self._active) >= self.side_min_active
        if self._side is not None:
            return  # one refill at a time: the next arrivals wait for it (at most a tick)
        if self._held is not None or not self._pending.empty():
            # A request that ends by its frame budget ends in a tick known since its admission.  Once that tick is queued the
            # slot can take its next tenant straight away: the prefill is queued behind the tick, and the snapshots (device-side
            # copies queued behind their ticks) still show the old tenant's last frames when the host gets to them.
            for slot, r in list(self._active.items()):
                if r.last_tick < self._tick_no and not r.cancelled:
                    r.retired = True
                    self._retiring.append(r)
                    del self._active[slot]
                    self._free.append(slot)
        while self._free and (self._held is not None or not self._pending.empty()):
            req, self._held = (self._held, None) if self._held is not None else (self._pending.get_nowait(), None)
            if req.cancelled:
                self._end(req)
                continue
            if req.prompt is None:
                try:
                    req.prompt = self.tts._get_prompt(req.text, req.voice)
                    if req.prompt.shape[1] + req.max_new_tokens + 2 > self.session.max_seq:
                        raise ValueError("prompt + max_new_tokens exceed max_seq_len")
                    if min(req.prompt.shape[1], self.prefill_chunk or req.prompt.shape[1]) > self.session.max_rows:
                        raise ValueError("prompt exceeds the session's prefill workspace")
                except Exception as e:  # bad request: answer it, keep serving
                    self._end(req, e)
                    continue
            need = req.prompt.shape[1] if side else min(req.prompt.shape[1], self.prefill_chunk or req.prompt.shape[1])
            if side and need > self.session.max_rows:
                side = False if not new else side  # a prompt too long for one side call goes in line, in chunks (alone)
                if new:
                    self._held = req
                    break
                need = min(req.prompt.shape[1], self.prefill_chunk or req.prompt.shape[1])
            if new and rows + need > self.session.max_rows:  # no room in this call: first in line next time
                self._held = req
                break
            rows += need
            req.slot = self._free.pop(0)
            new.append(req)
        if not new:
            return
        if side:
            try:
                h = self.session.side_park([r.prompt for r in new], [r.slot for r in new])  # queued behind the ticks so far
            except Exception as e:
                for r in new:
                    self._end(r, e)
                raise
            self._side = {"h": h, "reqs": new, "state": "parked"}
            return
        try:
            self._prefill(new)
        except Exception as e:  # these requests are in nobody's books yet: answer them here, then let the worker fail
            for r in new:
                self._end(r, e)
            raise
        self._enter(new)

    def _prefill(self, new: List[_Request]) -> None:
        if self.prefill_chunk:
            # long prompts (voice-clone speakers) enter in chunks; the slots already speaking get a tick in between
            def between():
                if self._active:
                    self._tick_and_snapshot()
                    self._consume_snapshots(keep=1)

            self.session.prefill_chunked([r.prompt for r in new], slots=[r.slot for r in new], stop_on_eos=True,
                                         chunk=self.prefill_chunk, between=between, defer_frame0=True)
        else:
            # frame 0 of the new slots comes out of the next tick's first frame (no separate tail for all slots)
            self.session.prefill([r.prompt for r in new], slots=[r.slot for r in new], stop_on_eos=True, defer_frame0=True)

    # ------------------------------------------------------------------ worker: ticks and snapshots
    def _tick_and_snapshot(self) -> None:
        """Queue one tick of frames and, behind it, a device-side copy of the output ring with an event: the host reads
        that copy later, while the following tick runs."""
        torch = self._torch
        self.session.decode(self.tick)
        s = self.session
        pcm = None
        streaming = [] if self.overlap_stream_codec else [r for r in self._active.values() if r.stream]
        if streaming:
            # the frames this tick gives slot b sit at ring positions [f0_b, f0_b + tick): f0_b follows from the tick count
            # alone while the request is alive (frames of a slot that has stopped are garbage here and never delivered)
            f0 = np.zeros(self.B, np.int64)
            for r in streaming:
                f0[r.slot] = (self._tick_no - r.first_tick) * self.tick
            # the pass runs over all slots; those without a stream (blocking requests, idle) decode garbage that nobody
            # reads, but their stream position advances too: restart them before it would reach the codec's capacity
            live = {r.slot for r in streaming}
            cap = int(self.tts.codec.c_cfg.max_positions)
            stale = [b for b in range(self.B) if b not in live and (self._codec_age[b] + 2) * 2 * self.tick > cap]
            if stale:
                self._stream_codec.reset_slots(stale)
                for b in stale:
                    self._codec_age[b] = 0
            for b in range(self.B):
                self._codec_age[b] += 1
            from ..engine import upload

            f0_d, = upload([f0], self.session.engine.device)
            idx = (f0_d[:, None] + torch.arange(self.tick, device="cuda")[None]).clamp_(max=self.max_frames - 1)
            nq = self.tts.config.num_codebooks
            chunk = s.codes[torch.arange(self.B, device="cuda")[:, None], idx][:, :, -nq:].contiguous()
            pcm = torch.empty(self.B, self.tick * 1920, dtype=torch.float32, device="cuda")
            self._stream_codec.decode_chunk(chunk, 0, self.tick, pcm, code_offset=0)
        snap = (s.codes.clone(), s.n_frames.clone(), s.done.clone(), torch.cuda.Event(), self._tick_no, pcm)
        snap[3].record(torch.cuda.current_stream())
        self._snaps.append(snap)
        self._tick_no += 1

    def _consume_snapshots(self, keep: int) -> None:
        """Read all but the ``keep`` newest snapshots (keep=1 in the steady state: the newest belongs to the tick that has
        only just been queued; its predecessor finished before that tick could start)."""
        torch = self._torch
        while len(self._snaps) > keep:
            codes_d, n_d, done_d, ev, tick_no, pcm_d = self._snaps.pop(0)
            # the host waits for the snapshot, then copies on the copy stream: a device-side wait would park a blocked barrier
            # packet in a second hardware queue for the whole tick, and the frame graphs' dependent launches get slower for it
            self._wait_event(ev)
            stream_pass = self._launch_stream_codec(codes_d, tick_no) if self.overlap_stream_codec else None
            with torch.cuda.stream(self._copy_stream):
                codes = codes_d.to("cpu", non_blocking=True)
                n_frames = n_d.to("cpu", non_blocking=True)
                done = done_d.to("cpu", non_blocking=True)
                pcm = pcm_d.to("cpu", non_blocking=True) if pcm_d is not None else None
            self._sync_copies()
            self._drain(codes.numpy(), n_frames.numpy(), done.numpy(), tick_no, None if pcm is None else pcm.numpy(), stream_pass)

    def _launch_stream_codec(self, codes_d, tick_no: int):
        """The codec pass of the streaming requests for tick ``tick_no``, on the codec stream, from the tick's snapshot of the
        output ring (the host has just seen that snapshot's event, so the next tick is running meanwhile).  Returns
        (pcm on the device, event) or None when no stream was alive in that tick."""
        torch = self._torch
        alive = [r for r in self._retiring + list(self._active.values())
                 if r.stream and r.first_tick <= tick_no <= r.last_tick and not r.closed and not r.stream_done]
        if not alive or self._stream_codec is None:
            return None
        from ..engine import upload

        with torch.cuda.stream(self._codec_stream):
            live = {r.slot for r in alive}
            cap = int(self.tts.codec.c_cfg.max_positions)
            restart = [r.slot for r in alive if r.first_tick == tick_no]  # new streams start at position 0 ...
            restart += [b for b in range(self.B) if b not in live and (self._codec_age[b] + 2) * 2 * self.tick > cap]  # ... idle slots before they overflow
            if restart:
                self._stream_codec.reset_slots(sorted(set(restart)))
                for b in restart:
                    self._codec_age[b] = 0
            for b in range(self.B):
                self._codec_age[b] += 1
            f0 = np.zeros(self.B, np.int64)
            for r in alive:
                f0[r.slot] = (tick_no - r.first_tick) * self.tick
            f0_d, = upload([f0], self.session.engine.device)
            idx = (f0_d[:, None] + torch.arange(self.tick, device="cuda")[None]).clamp_(max=self.max_frames - 1)
            nq = self.tts.config.num_codebooks
            chunk = codes_d[torch.arange(self.B, device="cuda")[:, None], idx][:, :, -nq:].contiguous()
            pcm = torch.empty(self.B, self.tick * 1920, dtype=torch.float32, device="cuda")
            self._stream_codec.decode_chunk(chunk, 0, self.tick, pcm, code_offset=0)
            ev = torch.cuda.Event()
            ev.record(self._codec_stream)
        return pcm, ev, codes_d  # (codes_d: kept alive until the pass has run)

    def _wait_event(self, ev) -> None:
        t = time.perf_counter()
        ev.synchronize()
        self._gpu_wait_s += time.perf_counter() - t

    def _sync_copies(self) -> None:
        t = time.perf_counter()
        self._copy_stream.synchronize()
        self._gpu_wait_s += time.perf_counter() - t

    def _drain(self, codes, n_frames, done, tick_no: int, pcm, stream_pass=None) -> None:
        stream_items = []  # overlapped codec pass of this tick: (request, pcm row, samples, last?) handed out by _deliver
        nq = self.tts.config.num_codebooks
        tc = self.tts.token_config

        def release(r):  # the request needs its slot no longer (a retired one has handed it over already)
            if r.retired:
                self._retiring.remove(r)
            else:
                del self._active[r.slot]
                self._free.append(r.slot)

        for r in self._retiring + list(self._active.values()):
            slot = r.slot
            if r.cancelled:  # the client is gone: free the slot now, whatever the snapshot shows
                release(r)
                r.pending = []
                self._end(r)
                continue
            if tick_no < r.first_tick:  # the snapshot predates this request: it shows the slot's previous tenant
                continue
            if r.retired and tick_no > r.last_tick:  # (cannot happen: a retired request ends with its last tick's snapshot)
                release(r)
                self._end(r, RuntimeError("scheduler lost the last frames of a request"))
                continue
            n = min(int(n_frames[slot]), r.max_new_tokens + 1)
            finished = (bool(done[slot]) and int(n_frames[slot]) > 0) or n >= r.max_new_tokens + 1
            if r.stream:
                # streaming requests decode every frame (__init__.py:88-92); this tick's PCM of the slot starts at its frame
                # r.emitted (== f0 of the tick: one codec frame per LM frame)
                k = n - r.emitted
                if self.overlap_stream_codec:
                    if k > 0 or finished:
                        assert k == 0 or (stream_pass is not None and r.emitted == (tick_no - r.first_tick) * self.tick), "stream bookkeeping out of step"
                        stream_items.append((r, slot, max(k, 0) * 1920, finished, r.emitted == 0))
                    r.emitted = n
                    if finished:
                        r.stream_done = True
                        release(r)  # the end marker follows the last chunk, in _deliver
                    continue
                if k > 0:
                    assert pcm is not None and r.emitted == (tick_no - r.first_tick) * self.tick, "stream bookkeeping out of step"
                    r.out.put(pcm[slot, : k * 1920].copy())
                    self._counts["frames_delivered"] += k
                r.emitted = n
                if finished:
                    release(r)
                    self._end(r)
                continue
            # blocking requests keep only frames whose slow id is a semantic token (generate_blocking, lm/generate.py:196-207)
            slow = codes[slot, r.emitted:n, 0]
            keep = (slow >= tc.semantic_start_id) & (slow <= tc.semantic_end_id)
            cols = codes[slot, r.emitted:n][keep][:, -nq:].astype(np.int32)
            r.emitted = n
            if cols.shape[0]:
                r.pending.append(cols)
            if finished:
                release(r)
                self._finished.append(r)
        if stream_items:
            pcm_d, ev, keep = stream_pass if stream_pass is not None else (None, None, None)
            urgent = any(it[4] for it in stream_items)  # a first chunk: handed out as soon as the pass is through
            self._deliveries.append((ev, pcm_d, [it[:4] for it in stream_items], urgent, keep))

    # ------------------------------------------------------------------ worker: codec passes and delivery
    def _codec_backlog(self) -> bool:
        return bool(self._finished) or any(j is not None for j in self._codec_jobs)

    def _decode_finished(self, force: bool) -> None:
        """Codec work of the blocking requests: a pool of CODEC_BATCH codec slots, each decoding one complete utterance
        CODEC_CHUNK frames per pass; a slot whose utterance is through takes the next one waiting (its stream restarts at
        position 0), so a pass stays as wide as the backlog allows instead of narrowing towards the longest utterance.  A pass
        is queued when the pool is full, when work has waited long enough — CODEC_WAIT frames of LM ticks while the LM batch is busy (wide
        passes cost half as much per frame as narrow ones), one tick otherwise — or when nothing else is going on; its PCM
        goes out chunk by chunk through _deliver."""
        from ..engine import MimiSession, upload

        torch = self._torch
        jobs = self._codec_jobs
        for b, j in enumerate(jobs):  # clients that went away
            if j is not None and j.req.cancelled:
                self._end(j.req)
                jobs[b] = None
        nq = self.tts.config.num_codebooks
        while self._finished and None in jobs:
            r = self._finished.pop(0)
            if r.cancelled:
                self._end(r)
                continue
            cols = np.concatenate(r.pending) if r.pending else np.zeros((0, nq), np.int32)
            r.pending = []
            if cols.shape[0] == 0:  # nothing to decode (every frame was non-semantic): just close the response, in order
                self._deliveries.append((None, None, [(r, 0, 0, True)], False, None))
                continue
            b = jobs.index(None)
            jobs[b] = _CodecJob(r, cols)
            self._codec_fresh.append(b)
        occupied = [b for b, j in enumerate(jobs) if j is not None]
        if not occupied:
            self._codec_wait = 0
            return
        self._codec_wait += 1
        busy = (not self._pending.empty()) or 4 * len(self._active) >= 3 * self.B
        if not (force or len(occupied) == len(jobs) or self._codec_wait >= (max(1, self.CODEC_WAIT // self.tick) if busy else 1)):
            return
        self._codec_wait = 0
        # these passes depend on nothing the frame graphs produce on the device (their codes come from the host): they run on
        # the codec stream, beside the ticks
        side = self._codec_stream if self.overlap_stream_codec else torch.cuda.current_stream()
        with torch.cuda.stream(side):
            self._decode_finished_pass(jobs, occupied, nq)

    def _decode_finished_pass(self, jobs, occupied, nq) -> None:
        from ..engine import MimiSession, upload

        torch = self._torch
        if self._batch_codec is None:
            self._batch_codec = MimiSession(self.tts.codec, max_batch=self.CODEC_BATCH, max_chunk_frames=self.CODEC_CHUNK, products=self.codec_products)
            self._batch_codec.reset()
        sess = self._batch_codec
        m = occupied[-1] + 1  # the pass runs over slots 0..m-1; a hole among them decodes zeros that nobody reads
        n = [min(self.CODEC_CHUNK, jobs[b].cols.shape[0] - jobs[b].done) if jobs[b] is not None else 0 for b in range(m)]
        nmax = max(n)
        # restart the streams of new utterances, and of holes before their position would reach the codec's capacity
        cap = int(self.tts.codec.c_cfg.max_positions)
        restart = sorted(set(self._codec_fresh) | {b for b in range(m) if jobs[b] is None and (self._codec_slot_age[b] + 2) * 2 * self.CODEC_CHUNK > cap})
        if restart:
            sess.reset_slots(restart)
            for b in restart:
                self._codec_slot_age[b] = 0
        self._codec_fresh = []
        grid = np.zeros((m, nmax, nq), np.int32)
        for b in range(m):
            if n[b]:
                grid[b, : n[b]] = jobs[b].cols[jobs[b].done: jobs[b].done + n[b]]
        codes, = upload([grid], self.session.engine.device)
        pcm = torch.empty(m, nmax * 1920, dtype=torch.float32, device="cuda")
        sess.decode_chunk(codes, 0, nmax, pcm, code_offset=0)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        items = []
        for b in range(m):
            self._codec_slot_age[b] += 1
            if not n[b]:
                continue
            j = jobs[b]
            j.done += n[b]
            fin = j.done >= j.cols.shape[0]
            items.append((j.req, b, n[b] * 1920, fin))
            if fin:
                jobs[b] = None
        self._deliveries.append((ev, pcm, items, False, None))

    def _deliver(self, wait: bool) -> None:
        """Hand finished codec passes to their requests, in order; ``wait``: block on the oldest one."""
        torch = self._torch
        while self._deliveries:
            ev, pcm, items, _, _ = self._deliveries[0]
            wait = wait or any(d[3] for d in self._deliveries)  # a stream's first chunk is somewhere in the line: do not dawdle
            if ev is not None:
                if not (wait or ev.query()):
                    return
                self._wait_event(ev)
                with torch.cuda.stream(self._copy_stream):
                    host = pcm.to("cpu", non_blocking=True)
                self._sync_copies()
                host = host.numpy()
            self._deliveries.pop(0)
            for r, b, n, fin in items:
                if n and not r.cancelled:
                    r.out.put(host[b, :n].copy())
                    self._counts["frames_delivered"] += n // 1920
                if fin:
                    self._end(r)
            wait = False

    # ------------------------------------------------------------------ worker: main loop
    def _run(self) -> None:
        torch = self._torch
        try:
            # the ticks are a chain of ~200 dependent launches per frame: their queue goes first wherever the command processor
            # has a choice (the codec passes beside them are ~100 launches per pass; measured with a kernel trace of
            # tools/bench_scheduler.py: no LM kernel gets longer beside a codec pass, the chain only loses time BETWEEN its kernels)
            prio = int(__import__("os").environ.get("SMOLTTS_TICK_PRIORITY", "-1"))
            compute = torch.cuda.Stream(priority=prio)
            self._copy_stream = torch.cuda.Stream()
            self._codec_stream = torch.cuda.Stream()
            self._side_stream = torch.cuda.Stream()
            with torch.cuda.stream(compute):
                while not self._stop.is_set():
                    self._admit()
                    if not self._active:
                        if self._side is not None:  # (its slots are the only ones taken: nothing to run beside)
                            self._side_advance()
                            continue
                        self._consume_snapshots(keep=0)
                        self._decode_finished(force=True)
                        if self._deliveries or self._codec_backlog():
                            self._deliver(wait=not self._codec_backlog())  # keep the passes coming while there is codec work
                            continue
                        self._wake.wait(timeout=0.05)  # idle: sleep until submit() (or close) without touching the queue
                        self._wake.clear()
                        continue
                    self._tick_and_snapshot()          # tick k and its snapshot are queued ...
                    self._decode_finished(force=False)
                    self._deliver(wait=False)
                    # ... while the host looks at what tick k-1 produced; unless a stream is still waiting for its first
                    # chunk: then this tick is read as soon as it is done (first-audio latency before pipelining)
                    first_chunk_due = any(r.stream and r.emitted == 0 for r in self._active.values())
                    if self.overlap_stream_codec:
                        # the codec pass of tick k-1 goes out now, beside tick k; reading tick k itself right away (keep=0) would
                        # leave the GPU idle until the next tick is queued -- with an arrival every other tick that was ~10 % of
                        # the wall time -- and would not bring the first chunk any earlier: it needs tick k finished either way
                        self._consume_snapshots(keep=1)
                        if any(d[3] for d in self._deliveries):
                            self._deliver(wait=False)  # a first chunk: wait for its pass (the running tick leaves the host slack)
                    else:
                        self._consume_snapshots(keep=0 if first_chunk_due else 1)
                    if self._side is not None and self._side["state"] == "parked":
                        self._side_advance()  # the tick before the park has been seen to finish: the side call runs beside this one
            self._fail_all(RuntimeError("scheduler closed"))  # requests still in flight when close() was called
        except Exception as e:  # engine failure: fail every waiter loudly
            self._fail_all(e)

    def _end(self, r: _Request, e: Optional[Exception] = None) -> None:
        """Queue the end marker of a request (exactly once)."""
        if not r.closed:
            r.closed = True
            self._counts["failed" if e is not None else ("cancelled" if r.cancelled else "completed")] += 1
            r.out.put(e)

    def stats(self) -> dict:
        """Counters since start (served by ``GET /v1/stats``): requests by outcome, audio frames handed to clients, ticks,
        slots in use, queue length."""
        up = time.time() - self._t0
        return dict(self._counts, ticks=self._tick_no, frames_per_tick=self.tick, slots=self.B, active=len(self._active),
                    queued=self._pending.qsize(), awaiting_codec=len(self._finished) + sum(1 for j in self._codec_jobs if j is not None), uptime_s=up, gpu_wait_s=self._gpu_wait_s,
                    delivered_frames_per_s=self._counts["frames_delivered"] / up if up > 0 else 0.0)

    def _fail_all(self, e: Exception) -> None:
        self._dead = e
        if self._held is not None:
            self._end(self._held, e)
            self._held = None
        side = self._side["reqs"] if self._side is not None else []
        self._side = None
        for r in list(self._active.values()) + side + self._retiring + self._finished + [j.req for j in self._codec_jobs if j is not None]:
            self._end(r, e)
        self._codec_jobs = [None] * len(self._codec_jobs)
        for d in self._deliveries:
            for r, _, _, _ in d[2]:
                self._end(r, e)
        self._active.clear()
        self._retiring = []
        self._finished = []
        self._deliveries = []
        while not self._pending.empty():
            self._end(self._pending.get_nowait(), e)
