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
for (M, N, K, epi, name) in [(32, 768, 768, E.EPI_RESID, "wo"), (32, 6144, 768, E.EPI_SWIGLU, "w13"), (32, 768, 3072, E.EPI_RESID, "w2"), (32, 16, 768, E.EPI_STORE, "1wg")]:
    x = torch.randn(M, K, device="cuda")
    gamma = torch.ones(K, device="cuda")
    x3, _, ssq = ops.x3_pack(x, gamma)
    w = ops.pack_weight(torch.randn(N, K) * 0.05)
    out = torch.zeros(M, N, device="cuda")
    x3o = ops.x3_alloc(M, N // 2) if epi == E.EPI_SWIGLU else None
    ea, ssqo = (ops.x3_alloc(M, N), torch.zeros(M, N // 16, device="cuda")) if epi == E.EPI_RESID else (None, None)
    for it in range(3):
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        ops.linear3(x3, w, M, N, K, epilogue=epi, ssq_in=None if epi == E.EPI_RESID else ssq, resid=out if epi == E.EPI_RESID else None,
                    out=None if epi == E.EPI_SWIGLU else out, x3_out=x3o, emit_a=ea, gamma_a=None, ssq_out=ssqo)
        torch.cuda.synchronize()
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(0))
    st = buf.cpu().view(16, 8, 2)
    t0, r0 = int(st[0, 0, 0]), int(st[0, 0, 1])
    print(f"== {name} M={M} N={N} K={K}")
    for wv in (0, 1, 7):
        print(f"   wave {wv}: " + "  ".join(f"s{k}={int(st[wv, k, 0]) - t0}cyc/{(int(st[wv, k, 1]) - r0) * 10}ns" for k in range(5)))
