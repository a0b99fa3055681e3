"""Do graph replays on two HIP streams overlap on this box? Diagnostic tool."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from smoltts_amd import engine as E, ops  # noqa: E402

dev = "cuda"
M, N, K = 16, 768, 768
x = torch.randn(M, K, device=dev)
x3, _, ssq = ops.x3_pack(x, torch.ones(K, device=dev))
w = ops.pack_weight(torch.randn(N, K) * 0.05)
outs = [torch.zeros(M, N, device=dev) for _ in range(2)]
n = 300
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
graphs = []
for i in range(2):
    with torch.cuda.stream(streams[i]):
        ops.linear3(x3, w, M, N, K, epilogue=E.EPI_STORE, ssq_in=ssq, out=outs[i])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[i]):
            for _ in range(n):
                ops.linear3(x3, w, M, N, K, epilogue=E.EPI_STORE, ssq_in=ssq, out=outs[i])
        graphs.append(g)
torch.cuda.synchronize()


def run(concurrent, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if concurrent:
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
            with torch.cuda.stream(streams[1]):
                graphs[1].replay()
        else:
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
                graphs[1].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


run(True); run(False)
a, b = run(False), run(True)
print(f"2 x {n} gemm3 launches (96 WGs each): same stream {a * 1e3:.2f} ms, two streams {b * 1e3:.2f} ms  -> overlap factor {a / b:.2f}")
