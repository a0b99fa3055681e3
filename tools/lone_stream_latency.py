"""Time to the first streamed chunk of a lone request through the scheduler (32-slot session, smoltts_byte_150m), repeated."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from smoltts_amd import SmolTTS  # noqa: E402
from smoltts_amd.codec.synthetic import synthetic_mimi_state  # noqa: E402
from smoltts_amd.config import GenerationSettings  # noqa: E402
from smoltts_amd.server.scheduler import BatchScheduler  # noqa: E402
from smoltts_amd.synthetic import named_config, synthetic_lm_state  # noqa: E402


def main():
    tick = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cfg = named_config("smoltts_byte_150m")
    tts = SmolTTS(state=synthetic_lm_state(cfg, seed=0), config=cfg, mimi_state=synthetic_mimi_state(seed=0))
    sched = BatchScheduler(tts, max_batch=32, frames_per_tick=tick, generation_settings=GenerationSettings.greedy(max_new_tokens=400))
    rng = np.random.default_rng(3)
    lat = []
    for i in range(12):
        text = "".join(chr(int(c)) for c in rng.integers(32, 127, size=int(rng.integers(40, 161))))
        time.sleep(0.1)
        t0 = time.perf_counter()
        it = sched.iter_chunks(sched.submit(text, "heart", stream=True, max_new_tokens=24))
        next(it)
        lat.append((time.perf_counter() - t0) * 1e3)
        for _ in it:
            pass
    print("first chunk ms per request:", " ".join(f"{x:.1f}" for x in lat), f"| p50 of the last 8: {np.median(lat[4:]):.1f}")
    sched.close()


if __name__ == "__main__":
    main()
