"""Mimi encoder timing (voice-clone enrolment): seconds of audio encoded per second, per signal length."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from smoltts_amd.codec.synthetic import synthetic_mimi_encoder_state, synthetic_mimi_state, synthetic_pcm  # noqa: E402
from smoltts_amd.engine import MimiEncoder  # noqa: E402

st = {**synthetic_mimi_state(seed=0), **synthetic_mimi_encoder_state(seed=0)}
enc = MimiEncoder(st, 8)
out = []
for secs in (1, 5, 10, 30):
    pcm = torch.from_numpy(synthetic_pcm(24000 * secs, 3)).cuda()
    enc.encode(pcm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        enc.encode(pcm)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    # algorithmic FLOPs: SEANet encoder convs + transformer + downsample + RVQ search (2 * MACs)
    macs = 0
    T, ch = 24000 * secs, 64
    macs += T * 7 * 64
    for r in (4, 5, 6, 8):
        macs += T * (3 * ch * ch // 2 + ch * ch // 2)
        T = -(-T // r)
        macs += T * 2 * r * ch * 2 * ch
        ch *= 2
    macs += T * 3 * 1024 * 512
    macs += T * 8 * (4 * 512 * 512 + 2 * 512 * 2048) + 8 * 2 * 64 * 8 * T * T // 2
    F = -(-T // 2)
    macs += F * (4 * 512 * 512 + 2 * 512 * 256 + 8 * 2048 * 256)
    out.append({"audio_s": secs, "ms": round(ms, 3), "x_realtime": round(secs / (ms / 1e3), 1), "tflops": round(2 * macs / (ms * 1e-3) / 1e12, 2)})
print(json.dumps({"mimi_encode": out}))
