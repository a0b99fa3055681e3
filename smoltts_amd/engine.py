"""ctypes binding of ``libsmoltts_hip.so`` (include/smoltts_hip.h) and the thin Python objects
around it.  PyTorch-ROCm is used only as a container for device memory and streams.

There is no CPU fallback: ``load_library`` raises when the HIP library has not been built, and
every wrapper raises ``SmolttsError`` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .config import NumericsMode, RQTransformerModelArgs, TokenConfig
from . import packing

LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libsmoltts_hip.so"
MAX_LAYERS, MAX_FAST_LAYERS, MIMI_MAX_LAYERS = 64, 16, 16


class SmolttsError(RuntimeError):
    pass


# ------------------------------------------------------------------------------- C structs
class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("attn_norm", "wqkv", "wo", "ffn_norm", "w13", "w2")]


class LMConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "n_layer", "n_head", "n_kv_head", "inter",
        "fast_dim", "n_fast_layer", "fast_n_head", "fast_n_kv_head", "fast_inter",
        "vocab_size", "codebook_size", "num_codebooks", "n_fast", "duplicate_code_0", "depthwise_wte",
        "has_fast_project_in", "embed_mask_mode", "semantic_start_id", "semantic_end_id", "im_end_id",
        "max_seq_len")] + [("norm_eps", C.c_float), ("weight_format", C.c_int32)]


class LMWeights(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "text_emb", "codebook_emb", "fast_emb", "norm", "head", "fast_norm", "fast_head",
        "fast_head_step_stride", "fast_proj_w", "fast_proj_b", "rope", "fast_rope")] + [
        ("layers", BlockWeights * MAX_LAYERS), ("fast_layers", BlockWeights * MAX_FAST_LAYERS)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("w_dev", C.c_void_p), ("w_is_fp32", C.c_int32), ("x_dev", C.c_void_p), ("ldx", C.c_int64),
        ("x_bstride", C.c_int64), ("rows_per_batch", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
        ("K", C.c_int32), ("prologue", C.c_int32), ("epilogue", C.c_int32), ("gamma_dev", C.c_void_p),
        ("eps", C.c_float), ("bias_dev", C.c_void_p), ("scale_dev", C.c_void_p), ("resid_dev", C.c_void_p),
        ("ldr", C.c_int64), ("r_bstride", C.c_int64), ("out_dev", C.c_void_p), ("ldo", C.c_int64), ("o_bstride", C.c_int64),
        ("elu_out", C.c_int32), ("raw_out_dev", C.c_void_p), ("raw_bstride", C.c_int64), ("rope_dev", C.c_void_p),
        ("row_pos_dev", C.c_void_p), ("row_slot_dev", C.c_void_p), ("k_cache_dev", C.c_void_p),
        ("v_cache_dev", C.c_void_p), ("n_q_heads", C.c_int32), ("n_kv_heads", C.c_int32), ("cache_len", C.c_int32),
        ("w3_dev", C.c_void_p), ("splitk_ws_dev", C.c_void_p), ("splitk_ws_floats", C.c_int64),
        ("beta_dev", C.c_void_p), ("ln_scratch_dev", C.c_void_p), ("k_cache3_dev", C.c_void_p), ("v_cache3_dev", C.c_void_p),
        ("b3_products", C.c_int32),
    ]


class MimiLayerWeights(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("ln1_w", "ln1_b", "wqkv", "wo", "ls1", "ln2_w", "ln2_b", "fc1", "fc2", "ls2",
                                          "wqkv3", "wo3", "fc13", "fc23")]


class MimiConv(C.Structure):
    _fields_ = [("w", C.c_uint64), ("b", C.c_uint64)] + [(n, C.c_int32) for n in ("cin", "cout", "k", "stride", "transposed")] + [
        ("_pad", C.c_int32), ("w3", C.c_uint64)]


class MimiConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_codebooks", "n_layers", "window", "max_positions")]


class MimiWeights(C.Structure):
    _fields_ = [("rvq_table", C.c_uint64), ("upsample_w", C.c_uint64), ("rope", C.c_uint64),
                ("layers", MimiLayerWeights * MIMI_MAX_LAYERS), ("convs", MimiConv * 14), ("final_w", C.c_uint64)]


class MimiEncConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_codebooks", "n_layers", "window", "max_positions", "extra_right")]


class MimiEncWeights(C.Structure):
    _fields_ = [("conv0_w", C.c_uint64), ("conv0_b", C.c_uint64), ("convs", MimiConv * 13),
                ("layers", MimiLayerWeights * MIMI_MAX_LAYERS), ("rope", C.c_uint64), ("downsample_w", C.c_uint64),
                ("in_proj", C.c_uint64 * 2), ("codebooks_t", C.c_uint64), ("codebooks", C.c_uint64), ("codebook_sq", C.c_uint64)]


PRO_NONE, PRO_RMSNORM, PRO_ELU, PRO_LAYERNORM = 0, 1, 2, 3
KV_FORMATS = {"fp32": 0, "bf16": 1}  # SMOLTTS_KV_*
EPI_STORE, EPI_RESID, EPI_SWIGLU, EPI_GELU, EPI_SCALE_RESID, EPI_QKV_ROPE = range(6)

_lib = None

_EXPORTS = [
    "smoltts_last_error", "smoltts_abi_version", "smoltts_engine_create", "smoltts_engine_destroy",
    "smoltts_session_slab_bytes", "smoltts_session_create", "smoltts_session_destroy", "smoltts_lm_prefill",
    "smoltts_lm_decode", "smoltts_session_outputs", "smoltts_mimi_create", "smoltts_mimi_destroy",
    "smoltts_mimi_slab_bytes", "smoltts_mimi_session_create", "smoltts_mimi_session_destroy", "smoltts_mimi_reset",
    "smoltts_mimi_decode_chunk", "smoltts_k_gemm", "smoltts_k_attention", "smoltts_k_embed", "smoltts_k_argmax",
    "smoltts_k_layernorm", "smoltts_k_gemm3", "smoltts_k_x3_pack",
    "smoltts_session_measure_duplicate", "smoltts_session_margin_at", "smoltts_session_drop_graph", "smoltts_session_set_frames_per_graph", "smoltts_engine_fast_qkv_bytes", "smoltts_engine_build_fast_qkv", "smoltts_session_set_option", "smoltts_gemm3_attn_fusable", "smoltts_lm_park_slots", "smoltts_lm_prefill_side", "smoltts_lm_start_slots", "smoltts_session_kv_cache", "smoltts_k_attention_split", "smoltts_k_attention_rows3",
    "smoltts_session_slab_bytes_kv", "smoltts_session_create_kv", "smoltts_k_attention_kv", "smoltts_session_set_sampling", "smoltts_k_sample",
    "smoltts_lm_prefill_chunk", "smoltts_lm_prefill_deferred", "smoltts_mimi_reset_slots", "smoltts_mimi_encoder_create", "smoltts_mimi_encoder_destroy", "smoltts_mimi_encode_frames",
    "smoltts_mimi_encode_workspace_bytes", "smoltts_mimi_encode", "smoltts_mimi_session_set_option",
]


def exported_symbols() -> List[str]:
    return list(_EXPORTS)


def load_library(path: Optional[Path] = None):
    """dlopen the in-tree HIP library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    override = os.environ.get("SMOLTTS_LIB")  # tools/ A/B runs: a variant built by `python -m smoltts_amd.build --variant ...`
    if path is None and override:
        print(f"[smoltts_amd] loading the library VARIANT {override} (SMOLTTS_LIB is set): not the product build", file=sys.stderr, flush=True)
    p = Path(path) if path is not None else (Path(override) if override else LIB_PATH)
    if not p.exists():
        raise SmolttsError(
            f"{p} not found: build it with `python -m smoltts_amd.build` (hipcc, gfx950). "
            "smoltts_amd has no CPU fallback.")
    lib = C.CDLL(str(p))
    lib.smoltts_last_error.restype = C.c_char_p
    lib.smoltts_session_slab_bytes.restype = C.c_size_t
    lib.smoltts_session_slab_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.smoltts_mimi_slab_bytes.restype = C.c_size_t
    lib.smoltts_mimi_slab_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.smoltts_session_slab_bytes_kv.restype = C.c_size_t
    lib.smoltts_session_slab_bytes_kv.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.smoltts_session_create_kv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.POINTER(C.c_void_p)]
    lib.smoltts_k_attention_kv.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.smoltts_engine_create.argtypes = [C.POINTER(LMConfig), C.POINTER(LMWeights), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.smoltts_engine_destroy.argtypes = [C.c_void_p]
    lib.smoltts_engine_destroy.restype = None
    lib.smoltts_session_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.smoltts_session_destroy.argtypes = [C.c_void_p]
    lib.smoltts_session_destroy.restype = None
    lib.smoltts_lm_prefill.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.smoltts_lm_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.smoltts_session_outputs.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4
    lib.smoltts_mimi_create.argtypes = [C.POINTER(MimiConfig), C.POINTER(MimiWeights), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.smoltts_mimi_destroy.argtypes = [C.c_void_p]
    lib.smoltts_mimi_destroy.restype = None
    lib.smoltts_mimi_session_create.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.smoltts_mimi_session_destroy.argtypes = [C.c_void_p]
    lib.smoltts_mimi_session_destroy.restype = None
    lib.smoltts_mimi_reset.argtypes = [C.c_void_p, C.c_void_p]
    lib.smoltts_mimi_decode_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    lib.smoltts_k_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    lib.smoltts_k_attention.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.smoltts_k_embed.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]
    lib.smoltts_k_argmax.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.smoltts_k_layernorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    lib.smoltts_session_set_sampling.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint64]
    lib.smoltts_k_sample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p]
    lib.smoltts_session_measure_duplicate.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.smoltts_session_margin_at.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.smoltts_session_drop_graph.argtypes = [C.c_void_p]
    lib.smoltts_session_set_frames_per_graph.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.smoltts_engine_fast_qkv_bytes.argtypes = [C.c_void_p]
    lib.smoltts_engine_fast_qkv_bytes.restype = C.c_size_t
    lib.smoltts_engine_build_fast_qkv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.smoltts_session_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.smoltts_k_attention_rows3.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p]
    lib.smoltts_k_attention_split.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.smoltts_mimi_reset_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.smoltts_lm_prefill_deferred.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                                C.c_int32, C.c_void_p]
    lib.smoltts_lm_prefill_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_void_p]
    lib.smoltts_lm_park_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.smoltts_lm_prefill_side.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.smoltts_lm_start_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.smoltts_mimi_encoder_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.smoltts_mimi_encoder_destroy.argtypes = [C.c_void_p]
    lib.smoltts_mimi_encoder_destroy.restype = None
    lib.smoltts_mimi_encode_frames.argtypes = [C.c_int32]
    lib.smoltts_mimi_encode_frames.restype = C.c_int32
    lib.smoltts_mimi_encode_workspace_bytes.argtypes = [C.c_void_p, C.c_int32]
    lib.smoltts_mimi_encode_workspace_bytes.restype = C.c_size_t
    lib.smoltts_mimi_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]
    if hasattr(lib, "smoltts_profile_begin"):  # diagnostic builds only (-DSMOLTTS_DEBUG_HOOKS)
        lib.smoltts_profile_begin.argtypes = [C.c_int32] * 4
        lib.smoltts_profile_end.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.smoltts_mimi_session_set_option.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    if lib.smoltts_abi_version() != 6:
        raise SmolttsError("libsmoltts_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load_library().smoltts_last_error().decode(errors="replace")
        raise SmolttsError(f"{what} failed ({status}): {msg}")


def _require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise SmolttsError("no HIP device visible; smoltts_amd runs on MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


_UPLOAD_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def upload_stream(device: torch.device) -> "torch.cuda.Stream":
    """The side stream of ``upload`` for this device, created (and the pinned allocator warmed) on first use; engines call
    this when they are built so that no request pays for it."""
    up = _UPLOAD_STREAMS.get(device.index)
    if up is None:
        up = _UPLOAD_STREAMS[device.index] = torch.cuda.Stream(device)
        # torch caches pinned blocks per power-of-two size class, and the first block of a class costs a hipHostMalloc (ms):
        # take a few of every class up to 1 MiB now
        warm = [[torch.empty(1 << k, dtype=torch.uint8).pin_memory() for _ in range(4)] for k in range(8, 21)]
        # ... and put one copy and one event through the stream: its hardware queue is only created by the first submission
        with torch.cuda.stream(up):
            warm[0][0].to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up)
        torch.cuda.current_stream(device).wait_event(ev)
        up.synchronize()
        del warm
    return up


def upload(arrays: Sequence[np.ndarray], device: torch.device) -> List[torch.Tensor]:
    """Host arrays -> device tensors usable on the current stream, without waiting for the work already queued on it.
    A plain ``tensor.to(device)`` from pageable memory is stream-ordered *and* blocks the host, i.e. it waits for everything
    the stream still has to do (a serving loop has a tick of frame graphs pending there); here the copies leave pinned memory on
    a side stream that is otherwise idle, and the current stream merely waits for their event."""
    cur = torch.cuda.current_stream(device)
    up = upload_stream(device)
    outs = []
    with torch.cuda.stream(up):
        for a in arrays:
            t = torch.from_numpy(np.ascontiguousarray(a)).pin_memory().to(device, non_blocking=True)
            t.record_stream(cur)  # allocated under the side stream, consumed on the current one
            outs.append(t)
        ev = torch.cuda.Event()
        ev.record(up)
    cur.wait_event(ev)
    return outs


def current_stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def dptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def _alloc_slab(nbytes: int, device) -> torch.Tensor:
    slab = torch.zeros(nbytes + 256, dtype=torch.uint8, device=device)
    shift = (-slab.data_ptr()) % 256
    return slab[shift: shift + nbytes]


# ------------------------------------------------------------------------------- LM engine
def lm_config_struct(cfg: RQTransformerModelArgs, tok: TokenConfig, numerics: NumericsMode, weight_format: int = 0) -> LMConfig:
    c = LMConfig()
    c.dim, c.n_layer, c.n_head, c.n_kv_head, c.inter = cfg.dim, cfg.n_layer, cfg.n_head, cfg.n_local_heads, cfg.intermediate_size
    c.fast_dim, c.n_fast_layer, c.fast_n_head = cfg.fast_dim, cfg.n_fast_layer, cfg.fast_n_head
    c.fast_n_kv_head, c.fast_inter = cfg.fast_n_local_heads, cfg.fast_intermediate_size
    c.vocab_size, c.codebook_size, c.num_codebooks = cfg.vocab_size, cfg.codebook_size, cfg.num_codebooks
    c.n_fast = cfg.max_fast_seqlen
    c.duplicate_code_0 = int(bool(cfg.duplicate_code_0))
    c.depthwise_wte = int(bool(cfg.depthwise_wte))
    c.has_fast_project_in = int(cfg.fast_dim != cfg.dim)
    c.embed_mask_mode = 0 if numerics.embed_mask == "torch" else 1
    c.semantic_start_id = tok.semantic_start_id
    c.semantic_end_id = tok.semantic_end_id if tok.semantic_end_id is not None else tok.semantic_start_id
    c.im_end_id = tok.im_end_id
    c.max_seq_len = cfg.max_seq_len
    c.norm_eps = cfg.norm_eps
    c.weight_format = int(weight_format)
    return c


def _fill_block(dst: BlockWeights, src: Dict[str, int]) -> None:
    for k, v in src.items():
        setattr(dst, k, v)


OPT_QKV_TABLE, OPT_COMMIT_PICKS, OPT_SPLIT_ATTN, OPT_STREAM_W, OPT_FUSE_DEPTH_ATTN, OPT_FUSE_PICK, OPT_FP8_PREFILL = 1, 2, 3, 4, 5, 6, 7  # include/smoltts_hip.h SMOLTTS_OPT_*


class LMEngine:
    """Immutable model on one GPU: packed weight arena + ``SmolttsEngine`` handle."""

    def __init__(self, cfg: RQTransformerModelArgs, state: Dict[str, torch.Tensor], token_config: TokenConfig,
                 numerics: Optional[NumericsMode] = None, arena: Optional[torch.Tensor] = None, offsets=None,
                 weight_format: str = "bf16", fast_qkv_table: Optional[bool] = None):
        """``weight_format="fp8"``: the Linears are stored as e4m3 with per-row scales (half the weight bytes);
        the model computed is exactly ``packing.fp8_reference_state`` of the checkpoint.
        ``fast_qkv_table`` (default on; ``SMOLTTS_QKV_TABLE=0`` switches the default off): build the engine's derived table of
        depth layer-0 q | k | v per fast-embedding row (``smoltts_engine_build_fast_qkv``: 7 launches fewer per frame)."""
        cfg.validate_for_engine()
        self.lib = load_library()
        self.device = _require_gpu()
        upload_stream(self.device)
        self.cfg, self.token_config = cfg, token_config
        self.numerics = numerics or NumericsMode.torch_reference()
        if arena is None:
            arena, offsets = packing.pack_lm(cfg, state, self.numerics, weight_format)
        self.offsets = offsets
        self.weight_format = "fp8" if offsets.get("weight_format", 0) else "bf16"
        self.arena = arena.to(self.device) if arena.device != self.device else arena
        self.c_cfg = lm_config_struct(cfg, token_config, self.numerics, offsets.get("weight_format", 0))
        w = LMWeights()
        for k in ("text_emb", "codebook_emb", "fast_emb", "norm", "head", "fast_norm", "fast_head",
                  "fast_head_step_stride", "fast_proj_w", "fast_proj_b", "rope", "fast_rope"):
            setattr(w, k, offsets[k])
        for i, b in enumerate(offsets["layers"]):
            _fill_block(w.layers[i], b)
        for i, b in enumerate(offsets["fast_layers"]):
            _fill_block(w.fast_layers[i], b)
        self.c_w = w
        h = C.c_void_p()
        check(self.lib.smoltts_engine_create(C.byref(self.c_cfg), C.byref(w), dptr(self.arena), self.arena.numel(), C.byref(h)),
              "smoltts_engine_create")
        self.handle = h
        if fast_qkv_table is None:
            fast_qkv_table = os.environ.get("SMOLTTS_QKV_TABLE", "1") != "0"
        self.fast_qkv = None
        need = self.lib.smoltts_engine_fast_qkv_bytes(self.handle) if fast_qkv_table else 0
        if need:
            self.fast_qkv = _alloc_slab(need, self.device)
            check(self.lib.smoltts_engine_build_fast_qkv(self.handle, dptr(self.fast_qkv), need, current_stream_ptr()),
                  "smoltts_engine_build_fast_qkv")

    @property
    def grid_height(self) -> int:
        return 1 + self.cfg.max_fast_seqlen

    def weight_bytes(self) -> int:
        return int(self.arena.numel())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.smoltts_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LMSession:
    """B utterance slots (KV caches + device-side frame loop state) inside one device slab."""

    def __init__(self, engine: LMEngine, max_batch: int, max_seq: Optional[int] = None, max_rows: int = 4096,
                 max_frames: int = 1025, kv_dtype: str = "fp32"):
        """``kv_dtype="bf16"``: the slow transformer's KV cache holds K (after RoPE) and V rounded to bf16 (half the
        attention stream; the oracle's ``kv_bf16=True`` is the same arithmetic).  Default fp32: K/V exactly as computed."""
        if kv_dtype not in KV_FORMATS:
            raise ValueError(f"kv_dtype must be one of {sorted(KV_FORMATS)}, got {kv_dtype!r}")
        self.kv_dtype = kv_dtype
        kvf = KV_FORMATS[kv_dtype]
        self.engine, self.lib = engine, engine.lib
        self.B = max_batch
        self.max_seq = max_seq or engine.cfg.max_seq_len
        self.max_rows = max(max_rows, max_batch)
        self.max_frames = max_frames
        self.H = engine.grid_height
        need = self.lib.smoltts_session_slab_bytes_kv(engine.handle, self.B, self.max_seq, self.max_rows, self.max_frames, kvf)
        if need == 0:
            raise SmolttsError("smoltts_session_slab_bytes returned 0 (bad sizes)")
        self.slab = _alloc_slab(need, engine.device)
        h = C.c_void_p()
        check(self.lib.smoltts_session_create_kv(engine.handle, dptr(self.slab), need, self.B, self.max_seq, self.max_rows,
                                                 self.max_frames, kvf, C.byref(h)), "smoltts_session_create")
        self.handle = h
        ptrs = [C.c_void_p() for _ in range(4)]
        check(self.lib.smoltts_session_outputs(h, *[C.byref(p) for p in ptrs]), "smoltts_session_outputs")
        base = self.slab.data_ptr()

        def view(p, nbytes, dtype, shape):
            o = p.value - base
            return self.slab[o: o + nbytes].view(dtype).view(*shape)

        self.codes = view(ptrs[0], self.B * self.max_frames * self.H * 4, torch.int32, (self.B, self.max_frames, self.H))
        self.n_frames = view(ptrs[1], self.B * 4, torch.int32, (self.B,))
        self.done = view(ptrs[2], self.B * 4, torch.int32, (self.B,))
        self.margin = view(ptrs[3], self.B * 4, torch.float32, (self.B,))
        mp = C.c_void_p()
        check(self.lib.smoltts_session_margin_at(h, C.byref(mp)), "smoltts_session_margin_at")
        self.margin_at = view(mp, self.B * 4, torch.int32, (self.B,))  # frame * 64 + step of each slot's smallest gap
        self._keep = None
        if os.environ.get("SMOLTTS_COMMIT_PICKS") == "0":  # A/B switches of tools/ (the ids are the same either way)
            self.use_commit_picks(False)
        if os.environ.get("SMOLTTS_SPLIT_ATTN") == "0":
            self.use_split_attention(False)
        if os.environ.get("SMOLTTS_FUSE_DEPTH_ATTN") == "0":
            self.use_fused_depth_attention(False)
        if os.environ.get("SMOLTTS_FUSE_PICK") == "0":
            self.use_fused_pick(False)
        if os.environ.get("SMOLTTS_FP8_PREFILL") == "1":  # (bench.py --fp8-prefill: not the parity path)
            self.use_fp8_prefill(True)
        if os.environ.get("SMOLTTS_STREAM_W") is not None:  # mask of SMOLTTS_STREAM_W_* bits
            check(self.lib.smoltts_session_set_option(self.handle, OPT_STREAM_W, int(os.environ["SMOLTTS_STREAM_W"])), "smoltts_session_set_option")

    def _rows(self, prompts, slots, pos0):
        """Prompt grids -> (grid rows, row slots, row positions on the device, last row per utterance, row count)."""
        cfg = self.engine.cfg
        cols, rslot, rpos, last = [], [], [], []
        n = 0
        for g, sl, p0 in zip(prompts, slots, pos0):
            g = np.asarray(g)
            if g.ndim != 2 or g.shape[0] != self.H or g.shape[1] < 1:
                raise ValueError(f"prompt grid must be ({self.H}, T>=1), got {g.shape}")
            T = g.shape[1]
            if p0 + T + 1 > self.max_seq:
                raise SmolttsError(f"prompt of {p0 + T} tokens does not fit max_seq={self.max_seq}")
            if g[0].min() < 0 or g[0].max() >= cfg.vocab_size or g[1:].min() < 0 or g[1:].max() >= cfg.codebook_size:
                raise ValueError("prompt ids out of range")
            cols.append(np.ascontiguousarray(g.T.astype(np.int32)))
            rslot.append(np.full(T, sl, np.int32))
            rpos.append(np.arange(p0, p0 + T, dtype=np.int32))
            n += T
            last.append(n - 1)
        if n > self.max_rows:
            raise SmolttsError(f"{n} prompt rows exceed the session's max_rows={self.max_rows}")
        grid_d, rslot_d, rpos_d = upload([np.concatenate(cols), np.concatenate(rslot), np.concatenate(rpos)], self.engine.device)
        return grid_d, rslot_d, rpos_d, last, n

    # ---- prompt prefill beside the decode frames (include/smoltts_hip.h at smoltts_lm_park_slots): three steps, the first and the
    #      last on the stream the frames run on, the middle one on any other stream once the first has run
    def side_park(self, prompts: Sequence[np.ndarray], slots: Sequence[int]):
        """Freeze the (idle) ``slots`` at their new prompts' last positions and upload the prompt rows; -> a handle for the two
        steps that follow.  Call on the frame stream."""
        slots = list(slots)
        if len(slots) != len(prompts) or len(set(slots)) != len(slots):
            raise ValueError("slots must be distinct and match prompts")
        grid_d, rslot_d, rpos_d, last, n = self._rows(prompts, slots, [0] * len(prompts))
        slots_h = (C.c_int32 * len(slots))(*slots)
        park_h = (C.c_int32 * len(slots))(*[int(np.asarray(g).shape[1]) - 1 for g in prompts])
        check(self.lib.smoltts_lm_park_slots(self.handle, slots_h, park_h, len(slots), current_stream_ptr()), "smoltts_lm_park_slots")
        parked = torch.cuda.Event()
        parked.record(torch.cuda.current_stream())
        return {"rows": (grid_d, rslot_d, rpos_d), "n": n, "slots": slots, "last": last, "parked": parked, "done": None}

    def side_run(self, h) -> None:
        """The prompts' KV rows, on the CURRENT stream (not the frames' one); the park must have run: the host waits for it here."""
        h["parked"].synchronize()
        grid_d, rslot_d, rpos_d = h["rows"]
        check(self.lib.smoltts_lm_prefill_side(self.handle, dptr(grid_d), dptr(rslot_d), dptr(rpos_d), h["n"], current_stream_ptr()),
              "smoltts_lm_prefill_side")
        h["done"] = torch.cuda.Event()
        h["done"].record(torch.cuda.current_stream())

    def side_start(self, h, stop_on_eos: bool = True) -> None:
        """Arm the slots (their frame 0 comes out of the next decode frame).  Call on the frame stream; waits (host) for the side call."""
        h["done"].synchronize()
        grid_d, _, rpos_d = h["rows"]
        slots_h = (C.c_int32 * len(h["slots"]))(*h["slots"])
        last_h = (C.c_int32 * len(h["slots"]))(*h["last"])
        check(self.lib.smoltts_lm_start_slots(self.handle, dptr(grid_d), dptr(rpos_d), slots_h, last_h, len(h["slots"]), int(stop_on_eos),
                                              current_stream_ptr()), "smoltts_lm_start_slots")
        self._keep = h["rows"]  # alive until the stream has consumed them

    def prefill(self, prompts: Sequence[np.ndarray], slots: Optional[Sequence[int]] = None, stop_on_eos: bool = True,
                pos0: Optional[Sequence[int]] = None, final: bool = True, defer_frame0: bool = False) -> None:
        """prompts: one ``(1 + n_fast, T_b)`` int grid per utterance; emits frame 0 of each slot.

        Chunked prefill: ``pos0[b]`` is the position of the first column of ``prompts[b]`` (its earlier columns
        went through previous calls with ``final=False``, which fill the KV cache only and leave the slot idle).
        ``defer_frame0``: no frame-0 tail here; the next ``decode`` call emits frame 0 as its first frame (serving loop)."""
        slots = list(range(len(prompts))) if slots is None else list(slots)
        if len(slots) != len(prompts) or len(set(slots)) != len(slots):
            raise ValueError("slots must be distinct and match prompts")
        pos0 = [0] * len(prompts) if pos0 is None else list(pos0)
        grid_d, rslot_d, rpos_d, last, n = self._rows(prompts, slots, pos0)
        slots_h = (C.c_int32 * len(slots))(*slots)
        last_h = (C.c_int32 * len(slots))(*last)
        self._keep = (grid_d, rslot_d, rpos_d)  # alive until the stream has consumed them
        if final and defer_frame0:
            check(self.lib.smoltts_lm_prefill_deferred(self.handle, dptr(grid_d), dptr(rslot_d), dptr(rpos_d), n, slots_h, last_h,
                                                       len(slots), int(stop_on_eos), current_stream_ptr()), "smoltts_lm_prefill_deferred")
        elif final:
            check(self.lib.smoltts_lm_prefill(self.handle, dptr(grid_d), dptr(rslot_d), dptr(rpos_d), n, slots_h, last_h,
                                              len(slots), int(stop_on_eos), current_stream_ptr()), "smoltts_lm_prefill")
        else:
            check(self.lib.smoltts_lm_prefill_chunk(self.handle, dptr(grid_d), dptr(rslot_d), dptr(rpos_d), n, slots_h, last_h,
                                                    len(slots), current_stream_ptr()), "smoltts_lm_prefill_chunk")

    def prefill_chunked(self, prompts: Sequence[np.ndarray], slots: Optional[Sequence[int]] = None, stop_on_eos: bool = True,
                        chunk: int = 128, between=None, defer_frame0: bool = False) -> None:
        """The same result as ``prefill`` with at most ``chunk`` columns per utterance per call; ``between()`` runs
        after every partial call (e.g. a few decode frames for the slots that are already speaking)."""
        slots = list(range(len(prompts))) if slots is None else list(slots)
        prompts = [np.asarray(g) for g in prompts]
        done = [0] * len(prompts)
        while True:
            part = [i for i, g in enumerate(prompts) if g.shape[1] - done[i] > chunk]
            if not part:
                break
            self.prefill([prompts[i][:, done[i]: done[i] + chunk] for i in part], [slots[i] for i in part], stop_on_eos,
                         pos0=[done[i] for i in part], final=False)
            for i in part:
                done[i] += chunk
            if between is not None:
                between()
        self.prefill([g[:, d:] for g, d in zip(prompts, done)], slots, stop_on_eos, pos0=done, final=True, defer_frame0=defer_frame0)

    def set_sampling(self, temp: float = 0.0, fast_temp: float = 0.0, min_p: float = 0.0, seed: int = 0) -> None:
        """temp / fast_temp <= 0: greedy (default). Takes effect from the next frame."""
        check(self.lib.smoltts_session_set_sampling(self.handle, float(temp), float(fast_temp), float(min_p), int(seed) & (2**64 - 1)),
              "smoltts_session_set_sampling")

    def measure_duplicate(self, code: int = -1, n_filter: int = 0) -> None:
        """Measurement aid (this session only): issue every launch of one kernel class twice; -1 switches it off."""
        check(self.lib.smoltts_session_measure_duplicate(self.handle, int(code), int(n_filter)), "smoltts_session_measure_duplicate")

    def decode(self, n_frames: int) -> None:
        check(self.lib.smoltts_lm_decode(self.handle, int(n_frames), current_stream_ptr()), "smoltts_lm_decode")

    def use_qkv_table(self, on: bool) -> None:
        """Depth layer-0 q | k | v from the engine's table (default where it exists) or through the wqkv GEMM (A/B, tests)."""
        check(self.lib.smoltts_session_set_option(self.handle, OPT_QKV_TABLE, 1 if on else 0), "smoltts_session_set_option")

    def use_split_attention(self, on: bool) -> None:
        """Slow attention of few rows with the keys of a (row, kv head) pair on two workgroups (default) or on one."""
        check(self.lib.smoltts_session_set_option(self.handle, OPT_SPLIT_ATTN, 1 if on else 0), "smoltts_session_set_option")

    def use_fused_depth_attention(self, on: bool) -> None:
        """Depth steps 1..: attention over the <= 8-entry cache inside the wo launch (default) or as a launch of its own."""
        check(self.lib.smoltts_session_set_option(self.handle, OPT_FUSE_DEPTH_ATTN, 1 if on else 0), "smoltts_session_set_option")

    def kv_cache(self):
        """(K, V) views of the slow transformer's cache: [n_layer, max_batch, n_kv_head, max_seq, 64] in the session's kv dtype (diagnostics)."""
        k, v, lb = C.c_void_p(), C.c_void_p(), C.c_uint64()
        check(self.lib.smoltts_session_kv_cache(self.handle, C.byref(k), C.byref(v), C.byref(lb)), "smoltts_session_kv_cache")
        cfg = self.engine.cfg
        n_layer, kvh = cfg.n_layer, cfg.n_local_heads
        dt = torch.float32 if lb.value == self.B * kvh * self.max_seq * 64 * 4 else torch.bfloat16
        base = self.slab.data_ptr()
        out = []
        for p in (k, v):
            o = p.value - base
            out.append(self.slab[o: o + n_layer * lb.value].view(dt).view(n_layer, self.B, kvh, self.max_seq, 64))
        return out

    def use_fp8_prefill(self, on: bool) -> None:
        """fp8-weight engines: prompt prefills of >= 256 rows on the fp8 x fp8 MFMA (BASELINE configs[4]'s fp8 MFMA prefill).  Faster
        first chunk; the prompt's KV rows carry the activations' fp8 rounding, so ids may leave the reference greedy decode."""
        check(self.lib.smoltts_session_set_option(self.handle, OPT_FP8_PREFILL, 1 if on else 0), "smoltts_session_set_option")

    def use_fused_pick(self, on: bool) -> None:
        """Greedy depth codes picked inside the next step's layer-0 attention + wo launch (default) or by a launch of their own."""
        check(self.lib.smoltts_session_set_option(self.handle, OPT_FUSE_PICK, 1 if on else 0), "smoltts_session_set_option")

    def use_commit_picks(self, on: bool) -> None:
        """The frame's slow token and last depth code picked inside the commit kernel (default) or in launches of their own."""
        check(self.lib.smoltts_session_set_option(self.handle, OPT_COMMIT_PICKS, 1 if on else 0), "smoltts_session_set_option")

    def set_frames_per_graph(self, n: int) -> None:
        """Frames per multi-frame graph (1 = single-frame graphs, 0 = follow the decode calls).  After a prefill and with
        n > 0 the graphs are captured now, on the current stream, instead of inside the first decode call."""
        check(self.lib.smoltts_session_set_frames_per_graph(self.handle, int(n), current_stream_ptr()), "smoltts_session_set_frames_per_graph")

    def fetch(self):
        """Synchronise and return (codes [B, max_frames, H] int32, n_frames [B], done [B], margin [B]) on the host."""
        torch.cuda.current_stream().synchronize()
        return (self.codes.cpu().numpy(), self.n_frames.cpu().numpy(), self.done.cpu().numpy(), self.margin.cpu().numpy())

    def close(self):
        if getattr(self, "handle", None):
            torch.cuda.synchronize()
            self.lib.smoltts_session_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------- Mimi engine
class MimiEngine:
    def __init__(self, state: Optional[Dict[str, torch.Tensor]], num_codebooks: int = 8, window: int = 0,
                 max_positions: int = 4096, arena: Optional[torch.Tensor] = None, offsets=None):
        self.lib = load_library()
        self.device = _require_gpu()
        if arena is None:
            arena, offsets = packing.pack_mimi(state, num_codebooks, max_positions)
        off = offsets
        max_positions = off["max_positions"]
        self.arena = arena.to(self.device)
        self.num_codebooks = num_codebooks
        cfg = MimiConfig(num_codebooks, off["n_layers"], window, max_positions)
        w = MimiWeights()
        w.rvq_table, w.upsample_w, w.rope = off["rvq_table"], off["upsample_w"], off["rope"]
        w.final_w = off["final_w"]
        for i, l in enumerate(off["layers"]):
            for k, v in l.items():
                setattr(w.layers[i], k, v)
        for i, cv in enumerate(off["convs"]):
            for k, v in cv.items():
                setattr(w.convs[i], k, v)
        self.c_cfg, self.c_w = cfg, w
        h = C.c_void_p()
        check(self.lib.smoltts_mimi_create(C.byref(cfg), C.byref(w), dptr(self.arena), self.arena.numel(), C.byref(h)),
              "smoltts_mimi_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.smoltts_mimi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MimiEncoder:
    """PCM -> RVQ codes (``MimiModel.encode``, codec/mimi.py:64-71) for voice-clone prompts.

    ``extra_right=False`` pads like the reference's MLX convs (everything on the left), ``True`` like
    ``transformers.MimiConv1d``; they agree for signals of whole frames (multiples of 1920 samples)."""

    def __init__(self, state: Optional[Dict[str, torch.Tensor]], num_codebooks: int = 8, window: int = 0, max_positions: int = 2048,
                 extra_right: bool = False, arena: Optional[torch.Tensor] = None, offsets=None):
        self.lib = load_library()
        self.device = _require_gpu()
        if arena is None:
            arena, offsets = packing.pack_mimi_encoder(state, num_codebooks, max_positions)
        off = offsets
        self.arena = arena.to(self.device)
        self.num_codebooks = off["num_codebooks"]
        cfg = MimiEncConfig(self.num_codebooks, off["n_layers"], window, off["max_positions"], int(extra_right))
        w = MimiEncWeights()
        for k in ("conv0_w", "conv0_b", "rope", "downsample_w", "codebooks_t", "codebooks", "codebook_sq"):
            setattr(w, k, off[k])
        w.in_proj[0], w.in_proj[1] = off["in_proj"]
        for i, l in enumerate(off["layers"]):
            for k, v in l.items():
                setattr(w.layers[i], k, v)
        for i, cv in enumerate(off["convs"]):
            for k, v in cv.items():
                setattr(w.convs[i], k, v)
        self.c_cfg, self.c_w = cfg, w
        h = C.c_void_p()
        check(self.lib.smoltts_mimi_encoder_create(C.byref(cfg), C.byref(w), dptr(self.arena), self.arena.numel(), C.byref(h)),
              "smoltts_mimi_encoder_create")
        self.handle = h
        self._ws = None

    def frames(self, n_samples: int) -> int:
        return int(self.lib.smoltts_mimi_encode_frames(n_samples))

    def encode(self, pcm, return_aux: bool = False):
        """pcm: 1-D float array/tensor of 24 kHz samples -> int32 device tensor (num_codebooks, frames)
        [, latents (frames, 512), squared-distance gaps (num_codebooks, frames)]."""
        x = torch.as_tensor(pcm, dtype=torch.float32).reshape(-1).to(self.device).contiguous()
        n = x.numel()
        if n == 0:
            raise SmolttsError("encode: empty signal")
        need = self.lib.smoltts_mimi_encode_workspace_bytes(self.handle, n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = _alloc_slab(need, self.device)
        F = self.frames(n)
        codes = torch.empty(self.num_codebooks, F, dtype=torch.int32, device=self.device)
        emb = torch.empty(F, 512, dtype=torch.float32, device=self.device) if return_aux else None
        gap = torch.empty(self.num_codebooks, F, dtype=torch.float32, device=self.device) if return_aux else None
        check(self.lib.smoltts_mimi_encode(self.handle, dptr(x), n, dptr(codes), dptr(emb) if return_aux else None,
                                           dptr(gap) if return_aux else None, dptr(self._ws), self._ws.numel(), current_stream_ptr()),
              "smoltts_mimi_encode")
        return (codes, emb, gap) if return_aux else codes

    def close(self):
        if getattr(self, "handle", None):
            self.lib.smoltts_mimi_encoder_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MimiSession:
    """Streaming Mimi decode state for ``max_batch`` slots; ``decode`` consumes frames chunk-wise."""

    SAMPLES_PER_FRAME = 1920
    OPT_STATELESS_UPSAMPLE = 1  # SMOLTTS_MIMI_OPT_STATELESS_UPSAMPLE
    OPT_PRODUCTS = 2  # SMOLTTS_MIMI_OPT_PRODUCTS

    def __init__(self, engine: MimiEngine, max_batch: int, max_chunk_frames: int = 8, stateless_upsample: bool = False,
                 products: int = 6):
        """``products`` = 3: the matrix-core kernels form three of the six bf16x3 products per operand pair (23 % faster chunks at a
        PCM RMS error of 7e-7 instead of 1e-7 against the fp32 oracle: include/smoltts_hip.h, SMOLTTS_MIMI_OPT_PRODUCTS).
        ``stateless_upsample``: every decode call up-samples its frames with no tap overlap carried in from the call before --
        the reference's ``decode_step`` (codec/mimi.py:73-77,101-104); off, chunked decode == batch decode."""
        self.engine, self.lib = engine, engine.lib
        self.B, self.chunk = max_batch, max_chunk_frames
        need = self.lib.smoltts_mimi_slab_bytes(engine.handle, max_batch, max_chunk_frames)
        if need == 0:
            raise SmolttsError("smoltts_mimi_slab_bytes returned 0 (bad sizes)")
        self.slab = _alloc_slab(need, engine.device)
        h = C.c_void_p()
        check(self.lib.smoltts_mimi_session_create(engine.handle, dptr(self.slab), need, max_batch, max_chunk_frames, C.byref(h)),
              "smoltts_mimi_session_create")
        self.handle = h
        if stateless_upsample:
            self.set_stateless_upsample(True)
        if products != 6:
            self.set_products(products)

    def set_products(self, n: int) -> None:
        check(self.lib.smoltts_mimi_session_set_option(self.handle, self.OPT_PRODUCTS, int(n)), "smoltts_mimi_session_set_option")

    def set_stateless_upsample(self, on: bool) -> None:
        check(self.lib.smoltts_mimi_session_set_option(self.handle, self.OPT_STATELESS_UPSAMPLE, int(bool(on))),
              "smoltts_mimi_session_set_option")

    def reset(self) -> None:
        check(self.lib.smoltts_mimi_reset(self.handle, current_stream_ptr()), "smoltts_mimi_reset")

    def reset_slots(self, slots: Sequence[int]) -> None:
        """Start new streams in the listed slots; the other slots' streams continue."""
        arr = (C.c_int32 * len(slots))(*slots)
        check(self.lib.smoltts_mimi_reset_slots(self.handle, arr, len(slots), current_stream_ptr()), "smoltts_mimi_reset_slots")

    def decode_chunk(self, codes: torch.Tensor, f0: int, n_frames: int, pcm: torch.Tensor, code_offset: int = 0) -> None:
        """codes: device int32 [batch, F, row] (row >= code_offset + num_codebooks); decodes frames
        [f0, f0+n_frames) of every slot into pcm[:, 1920*f0 : 1920*(f0+n_frames)]."""
        batch, F, row = codes.shape
        assert codes.dtype == torch.int32 and codes.is_contiguous() and pcm.dtype == torch.float32 and pcm.is_contiguous()
        assert batch <= self.B and n_frames <= self.chunk and f0 + n_frames <= F
        cptr = codes.data_ptr() + 4 * f0 * row
        pptr = pcm.data_ptr() + 4 * f0 * self.SAMPLES_PER_FRAME
        check(self.lib.smoltts_mimi_decode_chunk(self.handle, cptr, F * row, row, code_offset, batch, n_frames, pptr,
                                                 pcm.shape[1], current_stream_ptr()), "smoltts_mimi_decode_chunk")

    def decode(self, codes: torch.Tensor, code_offset: int = 0, reset: bool = True) -> torch.Tensor:
        """codes device int32 [batch, F, row] -> pcm [batch, 1920 F] (== MimiModel.decode)."""
        if reset:
            self.reset()
        batch, F, _ = codes.shape
        pcm = torch.empty(batch, F * self.SAMPLES_PER_FRAME, dtype=torch.float32, device=codes.device)
        for f0 in range(0, F, self.chunk):
            self.decode_chunk(codes, f0, min(self.chunk, F - f0), pcm, code_offset)
        return pcm

    def close(self):
        if getattr(self, "handle", None):
            torch.cuda.synchronize()
            self.lib.smoltts_mimi_session_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
