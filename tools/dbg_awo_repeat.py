"""Stress of the bit-identity claim of attn_wo_kernel (diagnostic): the fused launch against the two launches, many times, with
other kernels in between that leave different LDS / cache contents behind."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from smoltts_amd import engine as E, ops
E.load_library()
def bf16r(t): return t.to(torch.bfloat16).float()
junk = torch.randn(2048, 2048, device="cuda")
total_bad = 0
for (M, Hq, KV, pos) in [(32, 9, 3, 2), (32, 12, 4, 1), (32, 9, 3, 5), (33, 12, 4, 7), (5, 6, 2, 2), (19, 8, 2, 4), (9, 5, 5, 2), (64, 12, 4, 4)]:
    g = torch.Generator().manual_seed(M * 131 + Hq * 7 + pos)
    K = N = Hq * 64
    q = torch.randn(M, K, generator=g) * 1.5
    kc = torch.randn(M, KV, 8, 64, generator=g); vc = torch.randn(M, KV, 8, 64, generator=g)
    kc[:, :, pos + 1:] = float("nan"); vc[:, :, pos + 1:] = float("nan")
    w = bf16r(torch.randn(N, K, generator=g) * 0.04); r = torch.randn(M, N, generator=g)
    wt = ops.pack_weight(w); qd, kd, vd = q.cuda(), kc.cuda(), vc.cuda()
    rp = torch.full((M,), pos, dtype=torch.int32).cuda(); rs = torch.arange(M, dtype=torch.int32).cuda()
    first_f = first_u = None
    bad_f = bad_u = bad_x = 0
    for it in range(400):
        k = it % 5
        if k == 0: _ = junk @ junk
        elif k == 1: _ = torch.softmax(junk, -1)
        elif k == 2: _ = torch.sort(junk[:256], -1)
        elif k == 3: _ = (junk * float("nan")).sum()
        rd = r.cuda()
        ops.linear3(None, wt, M, N, K, epilogue=E.EPI_RESID, resid=rd, out=rd, attn_q=qd, attn_pos=pos, k_cache=kd, v_cache=vd, n_q_heads=Hq, n_kv_heads=KV, cache_len=8)
        if k == 4: _ = torch.cumsum(junk, 0)
        ax3 = ops.x3_alloc(M, K)
        a2 = ops.attention(qd, kd, vd, rp, rs, Hq, out_x3=ax3)
        rd2 = r.cuda()
        ops.linear3(ax3, wt, M, N, K, epilogue=E.EPI_RESID, resid=rd2, out=rd2)
        f, u = rd.cpu(), rd2.cpu()
        if first_f is None: first_f, first_u = f, u
        if not torch.equal(f, first_f):
            bad_f += 1
            if bad_f == 1:
                d = (f != first_f).nonzero(); print("  fused differs from its first run at", d[:8].tolist(), "count", len(d), "iteration", it)
        if not torch.equal(u, first_u):
            bad_u += 1
            if bad_u == 1:
                d = (u != first_u).nonzero(); print("  unfused differs from its first run at", d[:8].tolist(), "count", len(d), "iteration", it)
        if not torch.equal(f, u): bad_x += 1
    total_bad += bad_f + bad_u + bad_x
    print((M, Hq, KV, pos), "400 runs: fused unstable", bad_f, "unfused unstable", bad_u, "fused != unfused", bad_x)
print("TOTAL", total_bad)
