"""Sustained serving throughput of the continuous-batching scheduler (SURVEY.md §8d EOS variant): 128 requests with
frame budgets U{64..384} (seed 1) through 32 slots, smoltts_byte_150m synthetic weights, greedy, blocking responses.

    python tools/bench_scheduler.py [n_requests] [frames_per_tick] [stream|block] [pool_workers]

``pool_workers`` > 0 puts the same load through the multi-GPU front-end (server/pool.py) with that many worker processes,
all on GPU 0 here: what the relay between processes costs."""
import functools
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from bench import make_prompts  # noqa: E402,F401
from smoltts_amd import SmolTTS  # noqa: E402
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.config import GenerationSettings  # noqa: E402
from smoltts_amd.server.scheduler import BatchScheduler  # noqa: E402
from smoltts_amd.synthetic import named_config, synthetic_lm_state  # noqa: E402


def main():
    n_req = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    tick = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    streaming = len(sys.argv) > 3 and sys.argv[3] == "stream"
    n_pool = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rng = np.random.default_rng(1)
    budgets = rng.integers(64, 385, size=n_req)
    rng2 = np.random.default_rng(2)
    texts = ["".join(chr(int(c)) for c in rng2.integers(32, 127, size=int(rng2.integers(40, 161)))) for _ in range(n_req)]
    if n_pool:
        from smoltts_amd.server.pool import GpuPool, synthetic_scheduler

        sched = GpuPool(functools.partial(synthetic_scheduler, "smoltts_byte_150m", 0, 0, 32, tick, 400), devices=[0] * n_pool)
    else:
        cfg = named_config("smoltts_byte_150m")
        import os
        if os.environ.get("CODEC"):  # experiment: CODEC_BATCH,CODEC_CHUNK,CODEC_WAIT of the blocking codec passes
            BatchScheduler.CODEC_BATCH, BatchScheduler.CODEC_CHUNK, BatchScheduler.CODEC_WAIT = (int(x) for x in os.environ["CODEC"].split(","))
        cfg.max_seq_len = int(os.environ.get("MAX_SEQ", cfg.max_seq_len))  # experiment: the session's KV stride
        tts = SmolTTS(state=synthetic_lm_state(cfg, seed=0), config=cfg, mimi_state=synthetic_mimi_state(seed=0))
        slots = int(os.environ.get("SLOTS", 32))
        sched = BatchScheduler(tts, max_batch=slots, frames_per_tick=tick, generation_settings=GenerationSettings.greedy(max_new_tokens=400),
                               codec_products=int(os.environ.get("CODEC_PRODUCTS", 6)))  # experiment: 3 = SMOLTTS_MIMI_OPT_PRODUCTS
    for _ in range(max(n_pool, 1) * 2):
        sched.synthesize("warm up", max_new_tokens=8)
    if streaming:
        list(sched.iter_chunks(sched.submit("warm up the stream path", "heart", stream=True, max_new_tokens=8)))
    samples = [0] * n_req


    first_chunk_ms = []


    def worker(i):
        if streaming:
            t1 = time.perf_counter()
            n = 0
            for j, chunk in enumerate(sched.iter_chunks(sched.submit(texts[i], "heart", stream=True, max_new_tokens=int(budgets[i])))):
                if j == 0:
                    first_chunk_ms.append((time.perf_counter() - t1) * 1e3)
                n += chunk.shape[0]
            samples[i] = n
        else:
            samples[i] = sched.synthesize(texts[i], "heart", max_new_tokens=int(budgets[i])).shape[0]


    st0 = sched.stats() if not n_pool else {}
    if not n_pool:
        import torch as _t
        from smoltts_amd.engine import MimiSession
        _codec_ev = []
        _orig_dc = MimiSession.decode_chunk

        def _dc(self, *a, **k):
            e0, e1 = _t.cuda.Event(enable_timing=True), _t.cuda.Event(enable_timing=True)
            e0.record()
            try:
                return _orig_dc(self, *a, **k)
            finally:
                e1.record()
                _codec_ev.append((e0, e1))
        MimiSession.decode_chunk = _dc
    phase = {}
    if not n_pool:  # host time of the worker loop by phase
        def timed(name):
            f = getattr(sched, name)

            def g(*a, **k):
                t = time.perf_counter()
                try:
                    return f(*a, **k)
                finally:
                    phase[name] = phase.get(name, 0.0) + time.perf_counter() - t
            setattr(sched, name, g)
        import torch
        gpu_ev = {"decode": [], "prefill": [], "codec": []}

        def gpu_timed(obj, name, key):
            f = getattr(obj, name)

            def g(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                try:
                    return f(*a, **k)
                finally:
                    e1.record()
                    gpu_ev[key].append((e0, e1))
            setattr(obj, name, g)
        gpu_timed(sched.session, "decode", "decode")
        gpu_timed(sched.session, "prefill", "prefill")
        for name in ("_admit", "_tick_and_snapshot", "_decode_finished", "_deliver", "_consume_snapshots", "_drain"):
            timed(name)
    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_req)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    frames = sum(samples) // 1920
    print(f"{n_req} {'streaming' if streaming else 'blocking'} requests, {frames} frames of audio in {dt:.2f} s -> {frames / dt:.0f} frames/s "
          f"({frames / dt / 12.5:.0f}x real time), tick {tick}" + (f", pool of {n_pool} worker processes" if n_pool else "")
          + (f"; time to first chunk p50 {np.median(first_chunk_ms):.1f} ms (includes queueing for a slot)" if first_chunk_ms else ""))
    if not n_pool:
        st1 = sched.stats()
        ticks = st1["ticks"] - st0["ticks"]
        torch.cuda.synchronize()
        gpu_ev["codec"] = _codec_ev
        print("  GPU time (s): " + ", ".join(f"{k} {sum(a.elapsed_time(b) for a, b in v) / 1e3:.2f} ({len(v)} calls)" for k, v in gpu_ev.items() if v))
        print("  host time by phase (s): " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(phase.items(), key=lambda kv: -kv[1])))
        print(f"  worker: {ticks} ticks ({ticks * tick * sched.B} slot-frames for {frames} delivered), waited for the GPU {st1['gpu_wait_s'] - st0['gpu_wait_s']:.2f} s of {dt:.2f} s")
    sched.close()


if __name__ == "__main__":  # worker processes of the pool import this module again
    main()
