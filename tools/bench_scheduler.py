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
        tts = SmolTTS(state=synthetic_lm_state(cfg, seed=0), config=cfg, mimi_state=synthetic_mimi_state(seed=0))
        sched = BatchScheduler(tts, max_batch=32, frames_per_tick=tick, generation_settings=GenerationSettings.greedy(max_new_tokens=400))
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
    sched.close()


if __name__ == "__main__":  # worker processes of the pool import this module again
    main()
