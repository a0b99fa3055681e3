"""Per-launch time of the skinny GEMM for LM decode shapes, measured as a captured chain of
identical launches (graph replay, HIP events).  Diagnostic tool, not part of the product path.

usage: python tools/microbench_gemm.py [--reps 200]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch  # noqa: E402

from smoltts_amd import engine as E  # noqa: E402
from smoltts_amd import ops  # noqa: E402


def timed_graph(fn, reps, replays=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(replays):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * replays)  # us per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    print(torch.cuda.get_device_name(0))
    # a trivial kernel chain for the per-launch floor
    z = torch.zeros(64, device=dev)
    print(f"torch add_ (1 block) chain: {timed_graph(lambda: z.add_(1.0), args.reps):.2f} us/launch")
    rows = []
    for (M, N, K, pro, epi, name) in [
        (32, 6144, 768, E.PRO_RMSNORM, E.EPI_SWIGLU, "w13"),
        (16, 6144, 768, E.PRO_RMSNORM, E.EPI_SWIGLU, "w13 M16"),
        (1, 6144, 768, E.PRO_RMSNORM, E.EPI_SWIGLU, "w13 M1"),
        (32, 1280, 768, E.PRO_RMSNORM, E.EPI_STORE, "qkv-like (store)"),
        (32, 768, 768, E.PRO_NONE, E.EPI_RESID, "wo"),
        (32, 768, 3072, E.PRO_NONE, E.EPI_RESID, "w2"),
        (32, 2368, 768, E.PRO_RMSNORM, E.EPI_STORE, "head"),
        (32, 256, 768, E.PRO_RMSNORM, E.EPI_STORE, "N=256"),
        (32, 16, 768, E.PRO_RMSNORM, E.EPI_STORE, "N=16 (1 WG)"),
    ]:
        x = torch.randn(M, K, device=dev)
        # rotate over several weight copies so the stream is not served from L2 alone
        n_copies = 8
        ws = [ops.pack_weight(torch.randn(N, K) * 0.05) for _ in range(n_copies)]
        gamma = torch.ones(K, device=dev)
        out_cols = N // 2 if epi == E.EPI_SWIGLU else N
        out = torch.zeros(M, out_cols, device=dev)
        resid = out if epi == E.EPI_RESID else None
        state = {"i": 0}

        def fn():
            w = ws[state["i"] % n_copies]
            state["i"] += 1
            ops.linear(x, w, N, prologue=pro, epilogue=epi, gamma=gamma if pro == E.PRO_RMSNORM else None, resid=resid, out=out)

        us = timed_graph(fn, args.reps)
        wbytes = N * K * 2
        rows.append((name, M, N, K, us, wbytes / us / 1e3))
        print(f"{name:18s} M={M:3d} N={N:5d} K={K:5d}: {us:7.2f} us/launch  weights {wbytes / 1e6:6.2f} MB -> {wbytes / us / 1e3:7.1f} GB/s")


def stamps():
    """Cycle stamps of workgroup (0,0) of one w13 launch: where a workgroup's latency goes."""
    import ctypes

    if not hasattr(E.load_library(), "smoltts_debug_set_stamps"):
        raise SystemExit("cycle stamps need the hooks variant: SMOLTTS_LIB=$(python -m smoltts_amd.build --variant hooks | tail -1) python tools/microbench_gemm.py ...")
    lib = E.load_library()
    buf = torch.zeros(16 * 8 * 2, dtype=torch.int64, device="cuda")
    M, N, K = 32, 6144, 768
    x = torch.randn(M, K, device="cuda")
    w = ops.pack_weight(torch.randn(N, K) * 0.05)
    gamma = torch.ones(K, device="cuda")
    for it in range(3):
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
        ops.linear(x, w, N, prologue=E.PRO_RMSNORM, epilogue=E.EPI_SWIGLU, gamma=gamma)
        torch.cuda.synchronize()
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(0))
        st = buf.cpu().view(16, 8, 2)
        t0 = int(st[0, 0, 0]); r0 = int(st[0, 0, 1])
        print(f"-- launch {it}: stamps relative to wave0 start (shader cycles | 100MHz ticks)")
        for wv in (0, 1, 7):
            row = [(int(st[wv, k, 0]) - t0, int(st[wv, k, 1]) - r0) for k in range(6)]
            print(f"   wave {wv}: " + "  ".join(f"s{k}={c}|{rt}" for k, (c, rt) in enumerate(row)))
        cyc = int(st[0, 5, 0]) - t0; ticks = int(st[0, 5, 1]) - r0
        if ticks > 0:
            print(f"   wave0 total {cyc} cycles over {ticks * 10} ns -> clock {cyc / (ticks * 10) :.2f} GHz")


if __name__ == "__main__":
    stamps()
    main()


def main3(reps=200):
    """Same shapes on the bf16x3 (X3 operand) path."""
    print("-- gemm3 (bf16 MFMA, X3 operand)")
    dev = "cuda"
    for (M, N, K, epi, name) in [
        (32, 6144, 768, E.EPI_SWIGLU, "w13"), (16, 6144, 768, E.EPI_SWIGLU, "w13 M16"), (1, 6144, 768, E.EPI_SWIGLU, "w13 M1"),
        (32, 1280, 768, E.EPI_STORE, "qkv-like"), (32, 768, 768, E.EPI_RESID, "wo"), (32, 768, 3072, E.EPI_RESID, "w2"),
        (32, 2368, 768, E.EPI_STORE, "head"), (32, 16, 768, E.EPI_STORE, "N=16 (1 WG)"),
    ]:
        x = torch.randn(M, K, device=dev)
        gamma = torch.ones(K, device=dev)
        x3, _, ssq = ops.x3_pack(x, gamma)
        ws = [ops.pack_weight(torch.randn(N, K) * 0.05) for _ in range(8)]
        out = torch.zeros(M, N, device=dev)
        x3o = ops.x3_alloc(M, N // 2) if epi == E.EPI_SWIGLU else None
        ea, ssqo = (ops.x3_alloc(M, N), torch.zeros(M, N // 16, device=dev)) if epi == E.EPI_RESID else (None, None)
        st = {"i": 0}

        def fn():
            w = ws[st["i"] % 8]
            st["i"] += 1
            ops.linear3(x3, w, M, N, K, epilogue=epi, ssq_in=None if epi == E.EPI_RESID else ssq, resid=out if epi == E.EPI_RESID else None,
                        out=None if epi == E.EPI_SWIGLU else out, x3_out=x3o, emit_a=ea, gamma_a=gamma[:N] if ea is not None and N <= K else None, ssq_out=ssqo)

        us = timed_graph(fn, reps)
        print(f"{name:18s} M={M:3d} N={N:5d} K={K:5d}: {us:7.2f} us/launch  weights {N * K * 2 / 1e6:6.2f} MB -> {N * K * 2 / us / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main3()
