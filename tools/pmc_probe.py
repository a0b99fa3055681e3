"""Smallest programs for checking that rocprofv3 --pmc works on the box: argv[1] = torch | gemm3."""
import sys

import torch

sys.path.insert(0, ".")
if sys.argv[1] == "torch":
    a = torch.randn(1024, 1024, device="cuda")
    print(float((a @ a).sum()))
else:
    from smoltts_amd import engine, ops

    engine.load_library()
    x = torch.randn(32, 768).cuda()
    w = torch.randn(6144, 768) * 0.03
    x3, _, ssq = ops.x3_pack(x, torch.ones(768).cuda())
    h = ops.x3_alloc(32, 3072)
    for _ in range(4):
        ops.linear3(x3, ops.pack_weight(w), 32, 6144, 768, epilogue=engine.EPI_SWIGLU, ssq_in=ssq, x3_out=h)
    torch.cuda.synchronize()
    print("ok")
