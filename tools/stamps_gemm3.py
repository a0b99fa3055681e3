"""Where a gemm3 workgroup's time goes: in-kernel cycle stamps of workgroup (0,0). Diagnostic tool."""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import os  # noqa: E402

import torch  # noqa: E402

from smoltts_amd.build import build_library  # noqa: E402

os.environ["SMOLTTS_LIB"] = str(build_library(variant="hooks"))  # the cycle stamps exist only in the -DSMOLTTS_DEBUG_HOOKS variant
from smoltts_amd import engine as E, ops  # noqa: E402

lib = E.load_library()
buf = torch.zeros(16 * 8 * 2, dtype=torch.int64, device="cuda")
Hq, Hkv = 12, 4
cases = [(32, 768, 768, E.EPI_RESID, "wo"), (32, (Hq + 2 * Hkv) * 64, 768, E.EPI_QKV_ROPE, "wqkv"), (32, 6144, 768, E.EPI_SWIGLU, "w13"),
         (32, 768, 3072, E.EPI_RESID, "w2"), (32, 16, 768, E.EPI_STORE, "1wg")]
names = ["start", "loads issued", "MFMAs done", "partials in LDS", "barrier passed", "end"]
order = [0, 5, 1, 2, 3, 4]  # stamp slots in program order
for (M, N, K, epi, name) in cases:
    x = torch.randn(M, K, device="cuda")
    gamma = torch.ones(K, device="cuda")
    x3, _, ssq = ops.x3_pack(x, gamma)
    w = ops.pack_weight(torch.randn(N, K) * 0.05)
    out = torch.zeros(M, N, device="cuda")
    x3o = ops.x3_alloc(M, N // 2) if epi == E.EPI_SWIGLU else None
    ea, ssqo = (ops.x3_alloc(M, N), torch.zeros(M, N // 16, device="cuda")) if epi == E.EPI_RESID else (None, None)
    kw = {}
    if epi == E.EPI_QKV_ROPE:
        ang = torch.outer(torch.arange(512).float(), 1.0 / (100000 ** (torch.arange(0, 64, 2).float() / 64)))
        kw = dict(rope=torch.stack([ang.cos(), ang.sin()], -1).contiguous().cuda(), row_pos=torch.arange(300, 300 + M, dtype=torch.int32).cuda() % 512,
                  row_slot=torch.arange(M, dtype=torch.int32).cuda(), k_cache=torch.zeros(M, Hkv, 512, 64).cuda(), v_cache=torch.zeros(M, Hkv, 512, 64).cuda(),
                  n_q_heads=Hq, n_kv_heads=Hkv, cache_len=512)
        out = torch.zeros(M, Hq * 64, device="cuda")
    best = None
    for it in range(6):
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        ops.linear3(x3, w, M, N, K, epilogue=epi, ssq_in=None if epi == E.EPI_RESID else ssq, resid=out if epi == E.EPI_RESID else None,
                    out=None if epi == E.EPI_SWIGLU else out, x3_out=x3o, emit_a=ea, gamma_a=None, ssq_out=ssqo, **kw)
        torch.cuda.synchronize()
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(0))
        st = buf.cpu().view(16, 8, 2)
        r0 = int(st[0, 0, 1])
        tot = (int(st[0, 4, 1]) - r0) * 10
        if it >= 2 and (best is None or tot < best[0]):
            best = (tot, st.clone())
    tot, st = best
    r0 = int(st[0, 0, 1])
    print(f"== {name} M={M} N={N} K={K}: workgroup (0,0), wave 0, ns since its start (100 MHz clock, fastest of 4 warm launches):")
    print("   " + " | ".join(f"{names[i]} {(int(st[0, k, 1]) - r0) * 10}" for i, k in enumerate(order)))
    last = max((int(st[wv, 1, 1]) - r0) * 10 for wv in range(8) if int(st[wv, 1, 1]))
    print(f"   slowest wave's MFMAs done at {last} ns")
