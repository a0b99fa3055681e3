"""Operator-level wrappers over the ``smoltts_k_*`` C entry points (device tensors in/out).

These mirror the reference's building blocks one to one so the parity tests read like the
reference modules: ``linear`` = RMSNorm/ELU + nn.Linear + epilogue (modeling/model/rq_transformer.py
:535-613), ``attention`` = scaled_dot_product_attention over a KV cache, ``embed`` =
BaseTransformer.embed, ``argmax`` = greedy sampling, ``layernorm`` = nn.LayerNorm.
All of them run the HIP kernels; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import engine as E
from .packing import tile_t16x32, tile_w3


def pack_weight(w: torch.Tensor, fp32: bool = False) -> torch.Tensor:
    """Row-major [N, K] (CPU or GPU, any float dtype) -> T16x32 tiles on the current GPU."""
    flat = tile_t16x32(w.detach().float().cpu(), torch.float32 if fp32 else torch.bfloat16)
    return flat.cuda()


def pack_weight_w3(w: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 [N, K] -> bf16x3 piece tiles ("W3", include/smoltts_hip.h) on the current GPU."""
    return tile_w3(w.detach().float().cpu()).view(torch.uint8).cuda()


def linear(x: torch.Tensor, w_tiles: torch.Tensor, N: int, *, w_fp32: bool = False, prologue: int = E.PRO_NONE,
           epilogue: int = E.EPI_STORE, gamma: Optional[torch.Tensor] = None, eps: float = 1e-5,
           bias: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
           resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           M: Optional[int] = None, K: Optional[int] = None, ldx: Optional[int] = None, x_bstride: int = 0,
           rows_per_batch: int = 0, ldo: Optional[int] = None, o_bstride: int = 0, ldr: int = 0, r_bstride: int = 0,
           rope: Optional[torch.Tensor] = None, row_pos: Optional[torch.Tensor] = None,
           row_slot: Optional[torch.Tensor] = None, k_cache: Optional[torch.Tensor] = None,
           v_cache: Optional[torch.Tensor] = None, n_q_heads: int = 0, n_kv_heads: int = 0, cache_len: int = 0,
           elu_out: bool = False, raw_out: Optional[torch.Tensor] = None, raw_bstride: int = 0,
           w3: Optional[torch.Tensor] = None, splitk_ws: Optional[torch.Tensor] = None,
           beta: Optional[torch.Tensor] = None, ln_scratch: Optional[torch.Tensor] = None,
           k_cache3: Optional[torch.Tensor] = None, v_cache3: Optional[torch.Tensor] = None, b3_products: int = 6) -> torch.Tensor:
    """``w3`` (``pack_weight_w3`` of the same fp32 matrix): many-row calls run the bf16x3-split kernel (gemm_b3.hip).
    ``k_cache3`` / ``v_cache3`` (``kv3_cache``): EPI_QKV_ROPE also writes the K / V rows as bf16x3 pieces (``attention_rows3``)."""
    lib = E.load_library()
    M = x.shape[0] if M is None else M
    K = x.shape[1] if K is None else K
    out_cols = N // 2 if epilogue == E.EPI_SWIGLU else (n_q_heads * 64 if epilogue == E.EPI_QKV_ROPE else N)
    if out is None:
        out = torch.empty(M, out_cols, dtype=torch.float32, device=x.device)
    a = E.GemmArgs()
    a.w_dev, a.w_is_fp32, a.x_dev = E.dptr(w_tiles), int(w_fp32), E.dptr(x)
    a.ldx = x.stride(0) if ldx is None else ldx
    a.x_bstride, a.rows_per_batch, a.M, a.N, a.K = x_bstride, rows_per_batch, M, N, K
    a.prologue, a.epilogue, a.gamma_dev, a.eps = prologue, epilogue, E.dptr(gamma), eps
    a.bias_dev, a.scale_dev, a.resid_dev = E.dptr(bias), E.dptr(scale), E.dptr(resid)
    a.ldr, a.r_bstride = ldr, r_bstride
    a.out_dev = E.dptr(out)
    a.ldo = (out.stride(0) if out.dim() == 2 else out_cols) if ldo is None else ldo
    a.o_bstride = o_bstride
    a.elu_out, a.raw_out_dev, a.raw_bstride = int(elu_out), E.dptr(raw_out), raw_bstride
    a.rope_dev, a.row_pos_dev, a.row_slot_dev = E.dptr(rope), E.dptr(row_pos), E.dptr(row_slot)
    a.k_cache_dev, a.v_cache_dev = E.dptr(k_cache), E.dptr(v_cache)
    a.n_q_heads, a.n_kv_heads, a.cache_len = n_q_heads, n_kv_heads, cache_len
    a.w3_dev = E.dptr(w3)
    a.splitk_ws_dev, a.splitk_ws_floats = E.dptr(splitk_ws), (splitk_ws.numel() if splitk_ws is not None else 0)
    a.beta_dev, a.ln_scratch_dev = E.dptr(beta), E.dptr(ln_scratch)
    a.k_cache3_dev, a.v_cache3_dev = E.dptr(k_cache3), E.dptr(v_cache3)
    a.b3_products = b3_products
    E.check(lib.smoltts_k_gemm(C.byref(a), E.current_stream_ptr()), "smoltts_k_gemm")
    return out


def attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, row_pos: torch.Tensor,
              row_slot: torch.Tensor, n_q_heads: int, window: int = 0, out_x3: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [rows, Hq*64]; caches [slots, KV, cache_len, 64] fp32 or bf16; -> [rows, Hq*64] (and the X3 operand if given)."""
    lib = E.load_library()
    n_kv, cache_len = k_cache.shape[1], k_cache.shape[2]
    assert k_cache.dtype == v_cache.dtype and k_cache.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty_like(q)
    E.check(lib.smoltts_k_attention_kv(E.dptr(q), E.dptr(k_cache), E.dptr(v_cache), E.dptr(row_pos), E.dptr(row_slot),
                                       q.shape[0], n_q_heads, n_kv, cache_len, window, E.dptr(out), E.dptr(out_x3),
                                       1 if k_cache.dtype == torch.bfloat16 else 0, E.current_stream_ptr()), "smoltts_k_attention_kv")
    return out


def kv3_cache(slots: int, n_heads: int, cache_len: int, device="cuda") -> torch.Tensor:
    """A zero-filled bf16x3 piece cache (include/smoltts_hip.h SMOLTTS_KV3_BYTES): uint8 [slots, heads, ceil32(cache_len) * 384]."""
    return torch.zeros(slots, n_heads, (cache_len + 31) // 32 * 32 * 384, dtype=torch.uint8, device=device)


def kv3_decode(cache3: torch.Tensor, cache_len: int, is_v: bool) -> torch.Tensor:
    """The fp32 values [slots, heads, cache_len, 64] a piece cache holds (hi + mid + lo, summed smallest first: exact) -- tests."""
    S, H = cache3.shape[0], cache3.shape[1]
    cl = (cache_len + 31) // 32 * 32
    raw = cache3.cpu().view(torch.bfloat16).float()  # [S, H, cl * 192]
    if not is_v:  # [tile][chunk][piece][lane = 16 q + r][8]: K[16 tile + r][32 chunk + 8 q + j]
        v = raw.view(S, H, cl // 16, 2, 3, 4, 16, 8)
        v = (v[:, :, :, :, 2] + v[:, :, :, :, 1]) + v[:, :, :, :, 0]  # [S, H, tile, chunk, q, r, j]
        return v.permute(0, 1, 2, 5, 3, 4, 6).reshape(S, H, cl, 64)[:, :, :cache_len].contiguous()
    v = raw.view(S, H, cl // 32, 4, 3, 4, 16, 8)  # [pb][dim tile][piece][q][r][j]: V[32 pb + (j < 4 ? 4q + j : 16 + 4q + j - 4)][16 t + r]
    v = (v[:, :, :, :, 2] + v[:, :, :, :, 1]) + v[:, :, :, :, 0]  # [S, H, pb, t, q, r, j]
    v = v.view(S, H, cl // 32, 4, 4, 16, 2, 4)  # j = 4 * half + jj -> position 16 half + 4 q + jj
    return v.permute(0, 1, 2, 6, 4, 7, 3, 5).reshape(S, H, cl, 64)[:, :, :cache_len].contiguous()


def kv3_encode(values: torch.Tensor, is_v: bool) -> torch.Tensor:
    """fp32 [slots, heads, cache_len, 64] -> the piece cache bytes (uint8 [slots, heads, ceil32(cache_len) * 384]); tests."""
    S, H, L, _ = values.shape
    cl = (L + 31) // 32 * 32
    x = torch.zeros(S, H, cl, 64)
    x[:, :, :L] = values.float().cpu()
    hi = x.bfloat16()
    r1 = x - hi.float()
    mid = r1.bfloat16()
    lo = (r1 - mid.float()).bfloat16()
    pieces = torch.stack([hi, mid, lo], 0)  # [3, S, H, cl, 64]
    if not is_v:  # -> [S, H, tile, chunk, piece, q, r, j]
        v = pieces.view(3, S, H, cl // 16, 16, 2, 4, 8).permute(1, 2, 3, 5, 0, 6, 4, 7)
    else:  # position = 32 pb + 16 half + 4 q + jj, dim = 16 t + r -> [S, H, pb, t, piece, q, r, half, jj]
        v = pieces.view(3, S, H, cl // 32, 2, 4, 4, 4, 16).permute(1, 2, 3, 7, 0, 5, 8, 4, 6)
    return v.contiguous().view(torch.uint8).reshape(S, H, cl * 384)


def attention_rows3(q: torch.Tensor, k_cache3: torch.Tensor, v_cache3: torch.Tensor, row_pos: torch.Tensor, row_slot: torch.Tensor,
                    rows_per_slot: int, n_heads: int, cache_len: int, window: int = 0) -> torch.Tensor:
    """Attention of ``rows_per_slot`` (a multiple of 32) consecutive positions per slot over bf16x3 piece caches (attn_rows3_kernel)."""
    lib = E.load_library()
    out = torch.empty_like(q)
    E.check(lib.smoltts_k_attention_rows3(E.dptr(q), E.dptr(k_cache3), E.dptr(v_cache3), E.dptr(row_pos), E.dptr(row_slot), q.shape[0],
                                          rows_per_slot, n_heads, cache_len, window, E.dptr(out), E.current_stream_ptr()),
            "smoltts_k_attention_rows3")
    return out


SPLIT_PART_FLOATS, SPLIT_TICKETS = 128 * 2 * (4 * 64 + 8), 128  # csrc/common.h ATT_SPLIT_*


def attention_split(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, row_pos: torch.Tensor, row_slot: torch.Tensor,
                    n_q_heads: int, scratch, window: int = 0, out_x3: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``attention`` with the keys of every (row, kv head) pair on two workgroups (rows x kv heads <= 128, cache_len > 128).
    ``scratch`` = (part fp32 [SPLIT_PART_FLOATS], ticket int32 [SPLIT_TICKETS] zeroed once and then left alone)."""
    lib = E.load_library()
    n_kv, cache_len = k_cache.shape[1], k_cache.shape[2]
    part, ticket = scratch
    assert part.numel() >= SPLIT_PART_FLOATS and ticket.numel() >= SPLIT_TICKETS and ticket.dtype == torch.int32
    out = torch.empty_like(q)
    E.check(lib.smoltts_k_attention_split(E.dptr(q), E.dptr(k_cache), E.dptr(v_cache), E.dptr(row_pos), E.dptr(row_slot),
                                          q.shape[0], n_q_heads, n_kv, cache_len, window, E.dptr(out), E.dptr(out_x3),
                                          1 if k_cache.dtype == torch.bfloat16 else 0, E.dptr(part), E.dptr(ticket),
                                          E.current_stream_ptr()), "smoltts_k_attention_split")
    return out


def embed(cols: torch.Tensor, text_emb: torch.Tensor, cb_emb: torch.Tensor, codebook_size: int, cb_first_offset: int = 0,
          mask_mode: int = 0, sem_start: int = 320, sem_end: int = 2367) -> torch.Tensor:
    """cols int32 [rows, 1+n]; bf16 tables; -> fp32 [rows, dim]."""
    lib = E.load_library()
    rows, dim = cols.shape[0], text_emb.shape[1]
    x = torch.empty(rows, dim, dtype=torch.float32, device=cols.device)
    E.check(lib.smoltts_k_embed(E.dptr(cols), rows, cols.shape[1] - 1, E.dptr(text_emb), E.dptr(cb_emb), dim, codebook_size,
                                cb_first_offset, mask_mode, sem_start, sem_end, E.dptr(x), E.current_stream_ptr()),
            "smoltts_k_embed")
    return x


def argmax(logits: torch.Tensor, margin: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = E.load_library()
    ids = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    E.check(lib.smoltts_k_argmax(E.dptr(logits), logits.shape[0], logits.shape[1], logits.stride(0), E.dptr(ids), 1,
                                 E.dptr(margin), E.current_stream_ptr()), "smoltts_k_argmax")
    return ids


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    lib = E.load_library()
    out = torch.empty_like(x)
    E.check(lib.smoltts_k_layernorm(E.dptr(x), E.dptr(w), E.dptr(b), x.shape[0], x.shape[1], eps, E.dptr(out),
                                    E.current_stream_ptr()), "smoltts_k_layernorm")
    return out


# ------------------------------------------------------------------------------- X3 / bf16-MFMA path
def x3_bytes(rows: int, K: int) -> int:
    return (rows + 15) // 16 * 16 * K * 6


def x3_alloc(rows: int, K: int) -> torch.Tensor:
    return torch.zeros(x3_bytes(rows, K), dtype=torch.uint8, device="cuda")


def x3_to_float(buf: torch.Tensor, rows: int, K: int) -> torch.Tensor:
    """Decode an X3 operand buffer back to fp32 [rows, K] on the CPU (tests): hi + mid + lo."""
    R16 = (rows + 15) // 16 * 16
    t = buf.cpu()[: R16 * K * 6].view(torch.bfloat16).view(R16 // 16, K // 32, 3, 4, 16, 8)  # tile, chunk, piece, q, r, j
    f = t.float().sum(dim=2)  # tile, chunk, q, r, j  (exact: the pieces do not overlap)
    return f.permute(0, 3, 1, 2, 4).reshape(R16, K)[:rows].contiguous()


def x3_pack(x: torch.Tensor, gamma_a: Optional[torch.Tensor] = None, gamma_b: Optional[torch.Tensor] = None,
            two: bool = False):
    """fp32 rows on the GPU -> (X3 of x*gamma_a, X3 of x*gamma_b or None, ssq [rows, K/16])."""
    lib = E.load_library()
    rows, K = x.shape
    a = x3_alloc(rows, K)
    b = x3_alloc(rows, K) if two else None
    ssq = torch.zeros(rows, K // 16, dtype=torch.float32, device=x.device)
    lib.smoltts_k_x3_pack.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 6
    E.check(lib.smoltts_k_x3_pack(E.dptr(x), x.stride(0), rows, K, E.dptr(a), E.dptr(gamma_a), E.dptr(b), E.dptr(gamma_b),
                                  E.dptr(ssq), E.current_stream_ptr()), "smoltts_k_x3_pack")
    return a, b, ssq


class Gemm3Args(C.Structure):
    _fields_ = [
        ("w_dev", C.c_void_p), ("x3_dev", C.c_void_p), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("epilogue", C.c_int32), ("ssq_in_dev", C.c_void_p), ("eps", C.c_float), ("bias_dev", C.c_void_p),
        ("resid_dev", C.c_void_p), ("out_dev", C.c_void_p), ("ldo", C.c_int64), ("x3_out_dev", C.c_void_p),
        ("emit_a_dev", C.c_void_p), ("gamma_a_dev", C.c_void_p), ("emit_b_dev", C.c_void_p), ("gamma_b_dev", C.c_void_p),
        ("ssq_out_dev", C.c_void_p), ("rope_dev", C.c_void_p), ("row_pos_dev", C.c_void_p), ("row_slot_dev", C.c_void_p),
        ("k_cache_dev", C.c_void_p), ("v_cache_dev", C.c_void_p), ("n_q_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("cache_len", C.c_int32), ("w_format", C.c_int32), ("w_scale_dev", C.c_void_p), ("v_x3_dev", C.c_void_p),
        ("kv_format", C.c_int32),
        ("w_stream", C.c_int32),
        ("attn_q_dev", C.c_void_p), ("attn_pos", C.c_int32),
        ("cand_out_dev", C.c_void_p), ("pick", C.c_void_p), ("fp8_activations", C.c_int32),
    ]


def pack_weight_fp8(w: torch.Tensor):
    """Row-major [N, K] -> (e4m3 T16x32 tiles, fp32 row scales) on the current GPU, and the dequantised matrix."""
    from .packing import quantize_fp8_rows

    q, scale = quantize_fp8_rows(w.detach().float().cpu())
    return tile_t16x32(q, torch.float8_e4m3fn).view(torch.uint8).cuda(), scale.cuda(), q.float() * scale[:, None]


def linear3(x3: torch.Tensor, w_tiles: torch.Tensor, M: int, N: int, K: int, *, epilogue: int = E.EPI_STORE,
            ssq_in: Optional[torch.Tensor] = None, eps: float = 1e-5, bias: Optional[torch.Tensor] = None,
            resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, x3_out: Optional[torch.Tensor] = None,
            emit_a: Optional[torch.Tensor] = None, gamma_a: Optional[torch.Tensor] = None,
            emit_b: Optional[torch.Tensor] = None, gamma_b: Optional[torch.Tensor] = None,
            ssq_out: Optional[torch.Tensor] = None, rope=None, row_pos=None, row_slot=None, k_cache=None, v_cache=None,
            n_q_heads: int = 0, n_kv_heads: int = 0, cache_len: int = 0, w_scale: Optional[torch.Tensor] = None,
            v_x3: Optional[torch.Tensor] = None, kv_format: int = 0, w_stream: bool = False,
            attn_q: Optional[torch.Tensor] = None, attn_pos: int = 0, cand_out: Optional[torch.Tensor] = None,
            fp8_activations: bool = False):
    """The bf16-MFMA GEMM over an X3 operand; returns the fp32 ``out`` tensor (None for SWIGLU).
    ``w_scale`` given: ``w_tiles`` are e4m3 tiles (``pack_weight_fp8``).
    ``attn_q`` given (EPI_RESID): ``x3`` may be None -- the operand is the attention of ``attn_q`` over keys 0..``attn_pos`` of
    ``k_cache`` / ``v_cache`` (row r = slot r), worked out inside the launch (attn_wo_kernel)."""
    lib = E.load_library()
    lib.smoltts_k_gemm3.argtypes = [C.POINTER(Gemm3Args), C.c_void_p]
    if out is None and epilogue != E.EPI_SWIGLU:
        cols = n_q_heads * 64 if epilogue == E.EPI_QKV_ROPE else N
        out = torch.zeros(M, cols, dtype=torch.float32, device=w_tiles.device)
    a = Gemm3Args()
    a.w_dev, a.x3_dev, a.M, a.N, a.K, a.epilogue = E.dptr(w_tiles), E.dptr(x3), M, N, K, epilogue
    a.ssq_in_dev, a.eps, a.bias_dev, a.resid_dev = E.dptr(ssq_in), eps, E.dptr(bias), E.dptr(resid)
    a.out_dev = E.dptr(out)
    a.ldo = out.stride(0) if out is not None else 0
    a.x3_out_dev = E.dptr(x3_out)
    a.emit_a_dev, a.gamma_a_dev, a.emit_b_dev, a.gamma_b_dev = E.dptr(emit_a), E.dptr(gamma_a), E.dptr(emit_b), E.dptr(gamma_b)
    a.ssq_out_dev = E.dptr(ssq_out)
    a.rope_dev, a.row_pos_dev, a.row_slot_dev = E.dptr(rope), E.dptr(row_pos), E.dptr(row_slot)
    a.k_cache_dev, a.v_cache_dev = E.dptr(k_cache), E.dptr(v_cache)
    a.n_q_heads, a.n_kv_heads, a.cache_len = n_q_heads, n_kv_heads, cache_len
    a.w_format, a.w_scale_dev = (1, E.dptr(w_scale)) if w_scale is not None else (0, None)
    a.v_x3_dev = E.dptr(v_x3)
    a.kv_format = int(kv_format)
    a.w_stream = 1 if w_stream else 0
    a.attn_q_dev, a.attn_pos = E.dptr(attn_q), int(attn_pos)
    a.fp8_activations = 1 if fp8_activations else 0  # fp8 weights, M >= 256: fp8 x fp8 MFMA on the activation's hi piece (not the parity path)
    a.cand_out_dev = E.dptr(cand_out)  # EPI_STORE: per (row, 16-column tile) (max, first column of it as int bits, runner-up, -)
    E.check(lib.smoltts_k_gemm3(C.byref(a), E.current_stream_ptr()), "smoltts_k_gemm3")
    return out
