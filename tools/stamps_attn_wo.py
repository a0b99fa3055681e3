"""Where a workgroup of the fused attention + wo launch (gemm3.hip attn_wo_kernel) spends its time, beside the plain wo GEMM:
in-kernel stamps of workgroup (0,0) (hooks build variant).  Diagnostic tool; run on the GPU box: python tools/stamps_attn_wo.py"""
import ctypes
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from smoltts_amd.build import build_library  # noqa: E402

os.environ["SMOLTTS_LIB"] = str(build_library(variant="hooks"))
from smoltts_amd import engine as E, ops  # noqa: E402

lib = E.load_library()
buf = torch.zeros(16 * 8 * 2, dtype=torch.int64, device="cuda")
M, Hq, KV = 32, 12, 4
K = N = Hq * 64
w = ops.pack_weight(torch.randn(N, K) * 0.05)
q = torch.randn(M, K, device="cuda")
kc, vc = torch.randn(M, KV, 8, 64, device="cuda"), torch.randn(M, KV, 8, 64, device="cuda")
x3, _, _ = ops.x3_pack(torch.randn(M, K, device="cuda"))
flush = torch.zeros(int(os.environ.get("FLUSH_MB", "48")) << 18, dtype=torch.float32, device="cuda")  # 48 MB: out of the 8 x 4 MB L2s, still in the Infinity Cache (what a depth layer sees in situ); FLUSH_MB=512: from HBM


def run(name, slots, labels, **kw):
    best = None
    for it in range(8):
        out = torch.zeros(M, N, device="cuda")
        ea, ssqo = ops.x3_alloc(M, N), torch.zeros(M, N // 16, device="cuda")
        flush.add_(1.0)
        buf.zero_()
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        ops.linear3(kw.get("x3"), w, M, N, K, epilogue=E.EPI_RESID, resid=out, out=out, emit_a=ea, ssq_out=ssqo,
                    **{k: v for k, v in kw.items() if k != "x3"})
        torch.cuda.synchronize()
        lib.smoltts_debug_set_stamps(ctypes.c_void_p(0))
        st = buf.cpu().view(16, 8, 2).clone()
        tot = int(st[0, 4, 1]) - int(st[0, 0, 1])
        if it >= 2 and (best is None or tot < best[0]):
            best = (tot, st)
    st = best[1]
    r0 = int(st[0, 0, 1])
    print(f"== {name}: workgroup (0,0), ns since wave 0's start (100 MHz clock, fastest of 6 warm launches)")
    for wv in (0, 3, 4, 11) if "attention" in name else (0, 3, 4, 7):  # attn_wo: waves 0..3 attention + epilogue, 4..11 weights + MFMAs
        print(f"   wave {wv}: " + " | ".join(f"{lab} {(int(st[wv, k, 1]) - r0) * 10}" for lab, k in zip(labels, slots) if int(st[wv, k, 1])))


run("plain wo (X3 operand from memory)", [0, 5, 1, 2, 3, 4], ["start", "loads issued", "MFMAs done", "partials in LDS", "barrier passed", "end"], x3=x3)
for pos in (1, 4, 7):
    run(f"attention + wo, {pos + 1} keys", [0, 5, 6, 1, 2, 3, 4],
        ["start", "all loads issued", "phase A done", "MFMAs done", "partials in LDS", "barrier passed", "end"],
        attn_q=q, attn_pos=pos, k_cache=kc, v_cache=vc, n_q_heads=Hq, n_kv_heads=KV, cache_len=8)
