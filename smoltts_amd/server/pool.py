"""Request-level data parallelism for serving: one worker *process* per GPU behind one front-end.

Utterances share no state, so a node's GPUs are N independent replicas: every worker process sees exactly one GPU
(``HIP_VISIBLE_DEVICES``), loads its own copy of the weights, and runs its own ``BatchScheduler``; the front-end hands
each request to the worker with the fewest requests in flight and relays the audio.  There is no exchange between the
workers — nothing a collective could carry — and the front-end never touches a GPU.  ``GpuPool`` has the client
interface of ``BatchScheduler`` (submit / synthesize / iter_chunks / cancel / close), so the HTTP handlers do not care
which of the two they talk to.

The reference serves from a single process and a single device (server.py:48-59); this is the part of the scaling story
that the single Python host would otherwise cap.
"""
from __future__ import annotations

import itertools
import multiprocessing as mp
import os
import queue
import threading
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


@dataclass
class _PoolRequest:
    rid: int
    worker: int
    out: "queue.Queue" = field(default_factory=queue.Queue)  # np.ndarray chunks, then None (or an Exception)
    cancelled: bool = False
    closed: bool = False


def visible_device(index: int, inherited: Optional[str]) -> str:
    """The ``HIP_VISIBLE_DEVICES`` value of the worker for GPU ``index``: an entry of the mask this process runs under, if
    there is one (indices are relative to it), else the index itself."""
    if inherited:
        ids = [x.strip() for x in inherited.split(",") if x.strip()]
        if index >= len(ids):
            raise ValueError(f"device {index} is outside HIP_VISIBLE_DEVICES={inherited!r}")
        return ids[index]
    return str(index)


class _Sender:
    """The worker's end of its result pipe; several pump threads share it."""

    def __init__(self, conn):
        self.conn, self.lock = conn, threading.Lock()

    def put(self, msg) -> None:
        with self.lock:
            self.conn.send(msg)


def _worker_main(device: str, factory: Callable[[], object], req_q, res_conn) -> None:
    """Worker process: build the scheduler on its GPU, then serve messages until told to close.  ``factory()`` returns an
    object with the BatchScheduler client interface (it is called here, after the device mask is in place and before
    anything has touched the GPU)."""
    os.environ["HIP_VISIBLE_DEVICES"] = device
    res_q = _Sender(res_conn)
    try:
        sched = factory()
    except BaseException as e:  # the front-end must hear about a worker that cannot start
        res_q.put((None, "fatal", f"{type(e).__name__}: {e}"))
        return
    res_q.put((None, "ready", device))
    live: Dict[int, object] = {}
    lock = threading.Lock()

    def pump(rid: int, req) -> None:
        try:
            for chunk in sched.iter_chunks(req):
                res_q.put((rid, "chunk", np.ascontiguousarray(chunk, dtype=np.float32)))
            res_q.put((rid, "end", None))
        except Exception as e:
            res_q.put((rid, "error", (type(e).__name__, str(e))))
        finally:
            with lock:
                live.pop(rid, None)

    try:
        while True:
            msg = req_q.get()
            if msg[0] == "close":
                break
            if msg[0] == "submit":
                _, rid, text, voice, stream, max_new_tokens = msg
                req = sched.submit(text, voice, stream=stream, max_new_tokens=max_new_tokens)
                with lock:
                    live[rid] = req
                threading.Thread(target=pump, args=(rid, req), name=f"smoltts-pump-{rid}", daemon=True).start()
            elif msg[0] == "cancel":
                with lock:
                    req = live.get(msg[1])
                if req is not None:
                    sched.cancel(req)
    finally:
        try:
            sched.close(drain=True)  # a BatchScheduler finishes what it has accepted
        except TypeError:
            sched.close()


class GpuPool:
    def __init__(self, factory: Callable[[], object], devices: Sequence[int], start_method: str = "spawn", ready_timeout: float = 600.0,
                 respawn: bool = True):
        """``factory``: picklable, called once in every worker to build its scheduler.  ``devices``: GPU indices, one
        worker each (an index may repeat: two replicas on one GPU).  ``start_method``: "spawn" or "forkserver" — never
        "fork": a forked copy of a process that has used the GPU is not usable.  ``respawn``: a worker that dies is replaced
        (same GPU); its requests in flight are failed, later ones are served again by all workers."""
        if start_method not in ("spawn", "forkserver"):
            raise ValueError("start_method must be 'spawn' or 'forkserver'")
        if not devices:
            raise ValueError("no devices")
        self._ctx = mp.get_context(start_method)
        self._factory, self._ready_timeout, self._respawn = factory, ready_timeout, respawn
        inherited = os.environ.get("HIP_VISIBLE_DEVICES")
        self._devices = [visible_device(d, inherited) for d in devices]
        started = [self._start_worker(i) for i in range(len(devices))]
        self._procs = [p for p, _, _ in started]
        self._req_qs = [q for _, q, _ in started]
        self._res = [r for _, _, r in started]  # one result pipe per worker: no shared lock, EOF when a worker dies
        self._lock = threading.Lock()
        self._reqs: Dict[int, _PoolRequest] = {}
        self._load: List[int] = [0] * len(devices)
        self._dead: List[bool] = [False] * len(devices)  # set (under the lock) by a worker's dispatcher when its pipe ends
        self._restarts: List[int] = [0] * len(devices)
        self._ids = itertools.count()
        self._closing = False
        for i, conn in enumerate(self._res):  # all workers up (weights loaded, kernels resident) before the first request is taken
            why = self._await_ready(conn, i)
            if why is not None:
                self._kill()
                raise RuntimeError(why)
        self._threads = [threading.Thread(target=self._dispatch, args=(i,), name=f"smoltts-pool-dispatch-{i}", daemon=True)
                         for i in range(len(devices))]
        for t in self._threads:
            t.start()

    def _start_worker(self, w: int):
        q = self._ctx.Queue()
        r, wr = self._ctx.Pipe(duplex=False)
        p = self._ctx.Process(target=_worker_main, args=(self._devices[w], self._factory, q, wr), daemon=True, name=f"smoltts-gpu-worker-{w}")
        p.start()
        wr.close()  # the worker holds the write end now
        return p, q, r

    def _await_ready(self, conn, w: int) -> Optional[str]:
        """None once worker ``w`` has reported ready, else why it did not."""
        try:
            if not conn.poll(self._ready_timeout):
                return "GPU workers did not come up in time"
            _, kind, payload = conn.recv()
        except (EOFError, OSError):
            return f"GPU worker failed to start: worker {w} exited during start-up"
        return None if kind == "ready" else f"GPU worker failed to start: {payload}"

    # ------------------------------------------------------------------ client side (the BatchScheduler interface)
    def submit(self, text: str, voice: str = "heart", stream: bool = False, max_new_tokens: Optional[int] = None) -> _PoolRequest:
        with self._lock:
            if self._closing:
                raise RuntimeError("pool closed")
            alive = [i for i, p in enumerate(self._procs) if not self._dead[i] and p.is_alive()]
            if not alive:
                raise RuntimeError("no GPU worker is alive")
            w = min(alive, key=lambda i: self._load[i])
            req = _PoolRequest(next(self._ids), w)
            self._reqs[req.rid] = req
            self._load[w] += 1
        self._req_qs[w].put(("submit", req.rid, text, voice, stream, max_new_tokens))
        return req

    def synthesize(self, text: str, voice: str = "heart", max_new_tokens: Optional[int] = None) -> np.ndarray:
        return np.concatenate(list(self.iter_chunks(self.submit(text, voice, False, max_new_tokens))) or [np.zeros(0, np.float32)])

    def iter_chunks(self, req: _PoolRequest):
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

    def cancel(self, req: _PoolRequest) -> None:
        if not req.cancelled and not req.closed:
            req.cancelled = True
            self._req_qs[req.worker].put(("cancel", req.rid))

    def stats(self) -> dict:
        """Front-end view (``GET /v1/stats``): workers alive and requests in flight per worker."""
        with self._lock:
            return {"workers": len(self._procs), "alive": sum(1 for i, p in enumerate(self._procs) if not self._dead[i] and p.is_alive()),
                    "in_flight": list(self._load), "restarts": list(self._restarts)}

    def loads(self) -> List[int]:
        """Requests in flight per worker."""
        with self._lock:
            return list(self._load)

    def close(self) -> None:
        with self._lock:
            self._closing = True
        for q in self._req_qs:
            q.put(("close",))
        for p in self._procs:
            p.join(timeout=60)
        self._kill()
        for t in self._threads:  # every dispatcher sees the end of its pipe once its worker is gone
            t.join(timeout=10)
        self._fail_open(RuntimeError("pool closed"))

    # ------------------------------------------------------------------ front-end internals
    def _finish(self, req: _PoolRequest, end) -> None:
        with self._lock:
            if req.closed:
                return
            req.closed = True
            self._reqs.pop(req.rid, None)
            self._load[req.worker] -= 1
        req.out.put(end)

    def _fail_open(self, e: Exception, worker: Optional[int] = None) -> None:
        with self._lock:
            reqs = [r for r in self._reqs.values() if worker is None or r.worker == worker]
        for r in reqs:
            self._finish(r, e)

    def _dispatch(self, w: int) -> None:
        """Relay worker ``w``'s output to the waiting clients; the end of its pipe means the worker is gone."""
        conn = self._res[w]
        while True:
            try:
                rid, kind, payload = conn.recv()
            except (EOFError, OSError):
                with self._lock:
                    self._dead[w] = True  # from here on submit() avoids this worker; what it holds is failed just below
                if self._closing:
                    return
                self._procs[w].join(timeout=5)
                self._fail_open(RuntimeError(f"GPU worker {w} died (exit code {self._procs[w].exitcode})"), worker=w)
                conn = self._replace(w)
                if conn is None:
                    return
                continue
            with self._lock:
                req = self._reqs.get(rid)
            if req is None:
                continue
            if kind == "chunk":
                if not req.cancelled:
                    req.out.put(payload)
            elif kind == "end":
                self._finish(req, None)
            elif kind == "error":
                name, msg = payload  # a refused request (ValueError) stays one: the HTTP layer answers 400 for it, 500 for the rest
                self._finish(req, ValueError(msg) if name == "ValueError" else RuntimeError(f"{name}: {msg}"))

    def _replace(self, w: int):
        """Start a new worker in place of the dead one; its result pipe, or None (not wanted, closing, or it keeps dying)."""
        if not self._respawn or self._restarts[w] >= 3:
            return None
        self._restarts[w] += 1
        p, q, r = self._start_worker(w)
        if self._await_ready(r, w) is not None or self._closing:
            if p.is_alive():
                p.terminate()
            return None
        with self._lock:
            self._procs[w], self._req_qs[w], self._res[w] = p, q, r
            self._dead[w] = False
        return r

    def _kill(self) -> None:
        for p in self._procs:
            if p.is_alive():
                p.terminate()  # the exact children this pool started
        for p in self._procs:
            p.join(timeout=10)


# ---------------------------------------------------------------------- factories (picklable: module-level callables)
def scheduler_from_settings(settings):
    """What a worker of ``smoltts-server --gpus N`` builds: the model from the settings file (a ``ServerSettings`` or the
    dict it is made from), and its scheduler."""
    from .. import SmolTTS
    from .scheduler import BatchScheduler
    from .settings import ServerSettings

    st = settings if isinstance(settings, ServerSettings) else ServerSettings(**settings)
    model = SmolTTS(checkpoint_dir=str(st.get_checkpoint_dir()), mimi_checkpoint=st.mimi_checkpoint, weight_format=st.weight_format)
    return BatchScheduler(model, max_batch=st.max_batch, generation_settings=st.generation.to_settings(), codec_products=st.codec_products)


def synthetic_scheduler(model: str = "tiny", seed: int = 21, mimi_seed: int = 5, max_batch: int = 4, frames_per_tick: int = 2,
                        max_new_tokens: int = 64):
    """Seeded random weights at the named shapes (tests, rehearsals without a checkpoint); greedy."""
    from .. import SmolTTS
    from ..codec.synthetic import synthetic_mimi_state
    from ..config import GenerationSettings
    from ..synthetic import named_config, synthetic_lm_state
    from .scheduler import BatchScheduler

    cfg = named_config(model)
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=seed), config=cfg, mimi_state=synthetic_mimi_state(seed=mimi_seed))
    return BatchScheduler(tts, max_batch=max_batch, frames_per_tick=frames_per_tick,
                          generation_settings=GenerationSettings.greedy(max_new_tokens=max_new_tokens))
