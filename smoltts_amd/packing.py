"""Host-side weight packing: reference checkpoints -> the device arena the HIP kernels stream.

* ``tile_t16x32``      the MFMA-fragment tile order documented in include/smoltts_hip.h.
* ``pack_lm``          torch / MLX state dicts of the DualAR model (keys of
                       modeling/model/rq_transformer.py; fused ``wqkv`` or legacy ``wq/wk/wv``
                       (load_hook :528-533); ``fast_output.weight`` as (n, d, 2048) torch layout or
                       (n*2048, d) flattened MLX layout, train/convert_safetensors.py:10-15;
                       ``_orig_mod.`` prefixes stripped, train/state.py) -> bf16 arena + offsets.
* ``pack_mimi``        Hugging Face ``kyutai/mimi`` decoder-side keys (codec/mimi.py:107-156) ->
                       fp32 arena + offsets, convolutions rewritten as GEMMs.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .config import NumericsMode, RQTransformerModelArgs

ALIGN = 256


def tile_t16x32(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Row-major [N][K] -> flat tiles; N zero-padded to 16, K must be a multiple of 32."""
    assert w.dim() == 2
    N, K = w.shape
    if K % 32:
        raise ValueError(f"K={K} must be a multiple of 32")
    Np = (N + 15) // 16 * 16
    if Np != N:
        w = torch.cat([w, torch.zeros(Np - N, K, dtype=w.dtype)], dim=0)
    w = w.to(dtype)
    if dtype == torch.float8_e4m3fn:  # same lane order as bf16, one byte per element (512-byte tiles)
        w = w.view(torch.uint8)
    if dtype in (torch.bfloat16, torch.float8_e4m3fn):
        t = w.reshape(Np // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4)  # nt, kc, q, r, 8
    elif dtype == torch.float32:
        t = w.reshape(Np // 16, 16, K // 32, 4, 2, 4).permute(0, 2, 4, 3, 1, 5)  # nt, kc, h, q, r, 4
    else:
        raise ValueError(dtype)
    return t.contiguous().reshape(-1)


def split3_bf16(w: torch.Tensor):
    """fp32 -> (hi, mid, lo) bf16 with hi + mid + lo == w exactly (3 x 8 significand bits), round to nearest even:
    the split the kernels apply to activations (csrc/x3.h split3_pair)."""
    w = w.float()
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def tile_w3(w: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 [N][K] -> "W3" tiles (include/smoltts_hip.h): per 16-row x 32-k tile the three pieces' bf16 T16x32
    blocks (1 KiB each) one after the other.  N zero-padded to 16, K % 32 == 0."""
    N, K = w.shape
    Np = (N + 15) // 16 * 16
    parts = [tile_t16x32(p_, torch.bfloat16).reshape(Np // 16, K // 32, 512) for p_ in split3_bf16(w)]
    return torch.stack(parts, dim=2).contiguous().reshape(-1)  # nt, kc, piece, 512 bf16


FP8_MAX = 448.0  # largest finite e4m3 (OCP "fn") value


def quantize_fp8_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row-major [N][K] -> (e4m3 codes [N][K], fp32 scale [N]) with w ~= codes * scale[:, None]; the row
    maximum maps to 448 (round to nearest even, as ``Tensor.to(float8_e4m3fn)`` does)."""
    w = w.float()
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    return (w / scale[:, None]).to(torch.float8_e4m3fn), scale


def dequantize_fp8_rows(w: torch.Tensor) -> torch.Tensor:
    q, scale = quantize_fp8_rows(w)
    return q.float() * scale[:, None]


def fp8_block(w: torch.Tensor) -> torch.Tensor:
    """One Linear in SMOLTTS_W_FP8 form: the e4m3 tiles followed by the fp32 row scales (bytes)."""
    if w.shape[0] % 16:
        raise ValueError(f"fp8 weights need N % 16 == 0, got {tuple(w.shape)}")
    q, scale = quantize_fp8_rows(w)
    return torch.cat([tile_t16x32(q, torch.float8_e4m3fn).view(torch.uint8), scale.contiguous().view(torch.uint8)])


def untile_t16x32(flat: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """Inverse of ``tile_t16x32`` (tests)."""
    Np = (N + 15) // 16 * 16
    if flat.dtype == torch.bfloat16:
        t = flat.reshape(Np // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4)
    else:
        t = flat.reshape(Np // 16, K // 32, 2, 4, 16, 4).permute(0, 4, 1, 3, 2, 5)
    return t.reshape(Np, K)[:N].contiguous()


class ArenaBuilder:
    def __init__(self):
        self.chunks: List[torch.Tensor] = []
        self.size = 0

    def add(self, t: torch.Tensor) -> int:
        raw = t.contiguous().reshape(-1).view(torch.uint8)
        off = self.size
        self.chunks.append(raw)
        pad = (-raw.numel()) % ALIGN
        if pad:
            self.chunks.append(torch.zeros(pad, dtype=torch.uint8))
        self.size += raw.numel() + pad
        return off

    def finish(self) -> torch.Tensor:
        return torch.cat(self.chunks) if self.chunks else torch.zeros(0, dtype=torch.uint8)


def rope_table(n_pos: int, head_dim: int, base: float, bf16: bool) -> torch.Tensor:
    """(n_pos, head_dim/2, 2) fp32 [cos, sin]; restates precompute_freqs_cis
    (modeling/model/rq_transformer.py:616-624) including its bf16 rounding when ``bf16``."""
    freqs = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(n_pos), freqs)
    fc = torch.polar(torch.ones_like(ang), ang)
    cache = torch.stack([fc.real, fc.imag], dim=-1)
    if bf16:
        cache = cache.to(torch.bfloat16)
    return cache.float().contiguous()


class _Reads(dict):
    """A state dict that remembers which keys a packer read (``report["unused"]``: the keys it did not)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.read = set()

    def __getitem__(self, k):
        self.read.add(k)
        return super().__getitem__(k)

    def unused(self) -> List[str]:
        return sorted(k for k in self if k not in self.read)


def _clean(state: Dict[str, torch.Tensor]) -> "_Reads":
    return _Reads({k.replace("_orig_mod.", ""): v for k, v in state.items()})


# Keys of the Hugging Face ``kyutai/mimi`` state dict (transformers.MimiModel) that neither half of the codec reads, and why.
MIMI_IGNORED = (
    (".codebook.initialized", "training bookkeeping of the k-means initialisation (codec/rvq.py never reads it)"),
    ("quantizer.acoustic_residual_vector_quantizer.layers.", "acoustic codebooks beyond the num_codebooks - 1 the model emits (31 stored, 7 used: codec/rvq.py:118-131)"),
)


def classify_mimi_keys(state: Dict[str, torch.Tensor], num_codebooks: int = 8) -> Dict[str, object]:
    """Which keys of a ``kyutai/mimi`` checkpoint the decoder packer reads, which the encoder packer reads, which are
    ignored on purpose (with the reason) and which are unknown (must be empty for a checkpoint this package understands)."""
    rd, re_ = {}, {}
    pack_mimi(state, num_codebooks, max_positions=64, report=rd)
    pack_mimi_encoder(state, num_codebooks, max_positions=64, report=re_)
    dec, enc = set(state) - set(rd["unused"]), set(state) - set(re_["unused"])
    ignored, unknown = {}, []
    for k in sorted(set(state) - dec - enc):
        why = next((w for pat, w in MIMI_IGNORED if (k.endswith(pat) if pat.startswith(".") else k.startswith(pat))), None)
        if why is None:
            unknown.append(k)
        else:
            ignored[k] = why
    return {"decoder": sorted(dec), "encoder": sorted(enc), "ignored": ignored, "unknown": unknown}


def _wqkv(st, prefix):
    if prefix + "attention.wqkv.weight" in st:
        return st[prefix + "attention.wqkv.weight"]
    return torch.cat([st[prefix + f"attention.{n}.weight"] for n in ("wq", "wk", "wv")])


def fp8_reference_state(cfg: RQTransformerModelArgs, state: Dict[str, torch.Tensor]):
    """The model the fp8 engine computes, as an ordinary checkpoint: every Linear (wqkv, wo, w1, w3, w2, both
    heads, fast_project_in) replaced by its dequantised e4m3 values; embedding *lookups* stay bf16, so a tied
    head becomes an explicit ``output.weight``.  -> (config with tie_word_embeddings=False, state)."""
    import dataclasses

    st = {k: v.float() for k, v in _clean(state).items()}
    out = dict(st)
    for k, v in st.items():
        if k.endswith((".wqkv.weight", ".wq.weight", ".wk.weight", ".wv.weight", ".wo.weight", ".w1.weight", ".w2.weight", ".w3.weight")) \
                or k in ("output.weight", "fast_project_in.weight"):
            out[k] = dequantize_fp8_rows(v)
    if cfg.tie_word_embeddings:
        out["output.weight"] = dequantize_fp8_rows(st["embeddings.weight"])
    fo = st["fast_output.weight"]
    if fo.dim() == 3:  # (n, d, cs): rows of the GEMM are W[i].T
        n, d, cs = fo.shape
        out["fast_output.weight"] = dequantize_fp8_rows(fo.permute(0, 2, 1).reshape(n * cs, d)).reshape(n, cs, d).permute(0, 2, 1).contiguous()
    else:
        out["fast_output.weight"] = dequantize_fp8_rows(fo)
    return dataclasses.replace(cfg, tie_word_embeddings=False), out


def pack_lm(cfg: RQTransformerModelArgs, state: Dict[str, torch.Tensor], numerics: NumericsMode, weight_format: str = "bf16",
            report: Dict[str, object] = None):
    """-> (arena uint8 CPU tensor, offsets dict).  Embedding tables are bf16; the Linears are bf16 T16x32
    tiles, or with ``weight_format="fp8"`` e4m3 tiles + per-row scales (``fp8_block``).  Tensors may be stored in any
    float dtype (bf16 checkpoints, fp32 trainer states).  ``report`` (a dict) receives ``unused``: keys nothing read."""
    if weight_format not in ("bf16", "fp8"):
        raise ValueError(weight_format)
    st = _clean(state)
    ab = ArenaBuilder()
    bf = torch.bfloat16
    off: Dict[str, object] = {"weight_format": 1 if weight_format == "fp8" else 0}

    def mat(w):  # one Linear [N][K] in the engine's weight format
        return fp8_block(w) if weight_format == "fp8" else tile_t16x32(w, bf)

    def f32(k):
        return st[k].float()

    def block(prefix):
        w1, w3 = st[prefix + "feed_forward.w1.weight"], st[prefix + "feed_forward.w3.weight"]
        w13 = torch.stack([w1.float(), w3.float()], dim=1).reshape(2 * w1.shape[0], w1.shape[1])
        return {
            "attn_norm": ab.add(f32(prefix + "attention_norm.weight")),
            "wqkv": ab.add(mat(_wqkv(st, prefix).float())),
            "wo": ab.add(mat(f32(prefix + "attention.wo.weight"))),
            "ffn_norm": ab.add(f32(prefix + "ffn_norm.weight")),
            "w13": ab.add(mat(w13)),
            "w2": ab.add(mat(f32(prefix + "feed_forward.w2.weight"))),
        }

    emb = f32("embeddings.weight")
    off["text_emb"] = ab.add(emb.to(bf))
    off["codebook_emb"] = ab.add(f32("codebook_embeddings.weight").to(bf))
    off["fast_emb"] = ab.add(f32("fast_embeddings.weight").to(bf))
    off["norm"] = ab.add(f32("norm.weight"))
    head = emb if cfg.tie_word_embeddings else f32("output.weight")
    off["head"] = ab.add(mat(head))
    off["fast_norm"] = ab.add(f32("fast_norm.weight"))
    fo = f32("fast_output.weight")
    n_fast, cs, fd = cfg.max_fast_seqlen, cfg.codebook_size, cfg.fast_dim
    if cfg.depthwise_output:
        if fo.dim() == 3:  # torch (n, d, cs): logits_i = h @ W[i]  => rows = W[i].T
            fo = fo.permute(0, 2, 1).reshape(n_fast * cs, fd)
        elif fo.shape != (n_fast * cs, fd):
            raise ValueError(f"fast_output.weight has shape {tuple(fo.shape)}")
        off["fast_head_step_stride"] = cs
    else:
        if fo.shape != (cs, fd):
            raise ValueError(f"fast_output.weight has shape {tuple(fo.shape)}")
        off["fast_head_step_stride"] = 0
    if weight_format == "fp8" and cfg.depthwise_output:  # one block per depth step so that each carries its own scales
        off["fast_head"] = ab.add(torch.cat([fp8_block(fo[i * cs:(i + 1) * cs]) for i in range(n_fast)]))
    else:
        off["fast_head"] = ab.add(mat(fo))
    if cfg.fast_dim != cfg.dim:
        off["fast_proj_w"] = ab.add(mat(f32("fast_project_in.weight")))
        off["fast_proj_b"] = ab.add(f32("fast_project_in.bias"))
    else:
        off["fast_proj_w"] = off["fast_proj_b"] = 0
    off["rope"] = ab.add(rope_table(cfg.max_seq_len, cfg.head_dim, cfg.rope_base, numerics.rope_bf16))
    off["fast_rope"] = ab.add(rope_table(n_fast, cfg.fast_head_dim, cfg.rope_base, numerics.rope_bf16))
    off["layers"] = [block(f"layers.{i}.") for i in range(cfg.n_layer)]
    off["fast_layers"] = [block(f"fast_layers.{i}.") for i in range(cfg.n_fast_layer)]
    if report is not None:
        report["unused"] = st.unused()
    return ab.finish(), off


# ------------------------------------------------------------------------------------- Mimi
RATIOS = (8, 6, 5, 4)
_HALF_SPLIT_PERM = torch.tensor([j // 2 + 32 * (j % 2) for j in range(64)])  # new row 2j <- j, 2j+1 <- j+32


def _perm_heads(w: torch.Tensor, n_heads: int = 8) -> torch.Tensor:
    """Reorder the rows of a q/k projection so that the half-split RoPE pair (j, j+32) of every head
    becomes the interleaved pair (2j, 2j+1).  q.k dot products are invariant under a common
    permutation of head dims, so attention is unchanged (codec/transformer.py:63-67 uses
    nn.RoPE(traditional=False))."""
    d = w.shape[0] // n_heads
    idx = torch.cat([h * d + _HALF_SPLIT_PERM for h in range(n_heads)])
    return w[idx]


def conv_as_gemm(w: torch.Tensor, b: torch.Tensor, transposed: bool, stride: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """GEMM form [N][K] of a causal conv over channel-last rows (DESIGN.md 'convolutions as GEMMs').

    Conv1d (cout, cin, k), stride 1: row t of the output reads the k consecutive input rows ending
    at t: W'[co][j*cin + ci] = w[co][ci][j].
    ConvTranspose1d (cin, cout, k = 2*stride): output rows t*s + r, r < s, read input rows (t-1, t):
    W'[r*cout + co][ci] = w[ci][co][r + s] and W'[r*cout + co][cin + ci] = w[ci][co][r]."""
    if not transposed:
        cout, cin, k = w.shape
        return w.permute(0, 2, 1).reshape(cout, k * cin).contiguous(), b.clone()
    cin, cout, k = w.shape
    s = stride
    assert k == 2 * s
    prev = w[:, :, s:].permute(2, 1, 0)  # r, co, ci   (input row t-1)
    cur = w[:, :, :s].permute(2, 1, 0)   # r, co, ci   (input row t)
    return torch.cat([prev, cur], dim=2).reshape(s * cout, 2 * cin).contiguous(), b.repeat(s)


def mimi_conv_specs():
    """(hf key, cin, cout, k, stride, transposed) in execution order (codec/seanet.py:99-139)."""
    specs = [("0", 512, 1024, 7, 1, False)]
    ch, li = 1024, 1
    for r in RATIOS:
        specs.append((str(li + 1), ch, ch // 2, 2 * r, r, True))
        specs.append((f"{li + 2}.block.1", ch // 2, ch // 4, 3, 1, False))
        specs.append((f"{li + 2}.block.3", ch // 4, ch // 2, 1, 1, False))
        ch //= 2
        li += 3
    specs.append(("14", 64, 1, 3, 1, False))
    return specs


def pack_mimi(state: Dict[str, torch.Tensor], num_codebooks: int = 8, max_positions: int = 4096, report: Dict[str, object] = None):
    st = _Reads({k: v.float() for k, v in state.items()})
    ab = ArenaBuilder()
    f32 = torch.float32
    off: Dict[str, object] = {}
    # RVQ: fold embed = embed_sum / max(usage, eps) (rvq.py:41-48) and the group's 1x1 output_proj
    tables = []
    for q in range(num_codebooks):
        grp = "semantic" if q == 0 else "acoustic"
        p = f"quantizer.{grp}_residual_vector_quantizer."
        li = 0 if q == 0 else q - 1
        emb = st[p + f"layers.{li}.codebook.embed_sum"] / torch.clamp(st[p + f"layers.{li}.codebook.cluster_usage"], min=1e-5)[:, None]
        proj = st[p + "output_proj.weight"][:, :, 0]  # (512, 256)
        tables.append((emb.double() @ proj.double().T).float())
    off["rvq_table"] = ab.add(torch.stack(tables))
    off["upsample_w"] = ab.add(st["upsample.conv.weight"][:, 0, :].T.contiguous())  # [4][512]
    off["rope"] = ab.add(rope_table(max_positions, 64, 10000.0, bf16=False))
    n_layers = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("decoder_transformer.layers."))
    layers = []
    for li in range(n_layers):
        p = f"decoder_transformer.layers.{li}."
        wq, wk, wv = (st[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v"))
        wqkv = torch.cat([_perm_heads(wq), _perm_heads(wk), wv])
        layers.append({
            "ln1_w": ab.add(st[p + "input_layernorm.weight"]), "ln1_b": ab.add(st[p + "input_layernorm.bias"]),
            "wqkv": ab.add(tile_t16x32(wqkv, f32)),
            "wo": ab.add(tile_t16x32(st[p + "self_attn.o_proj.weight"], f32)),
            "ls1": ab.add(st[p + "self_attn_layer_scale.scale"]),
            "ln2_w": ab.add(st[p + "post_attention_layernorm.weight"]), "ln2_b": ab.add(st[p + "post_attention_layernorm.bias"]),
            "fc1": ab.add(tile_t16x32(st[p + "mlp.fc1.weight"], f32)),
            "fc2": ab.add(tile_t16x32(st[p + "mlp.fc2.weight"], f32)),
            "ls2": ab.add(st[p + "mlp_layer_scale.scale"]),
            # the same matrices as bf16x3 piece tiles: the many-row calls (chunked decode) run on the bf16 matrix cores
            "wqkv3": ab.add(tile_w3(wqkv)), "wo3": ab.add(tile_w3(st[p + "self_attn.o_proj.weight"])),
            "fc13": ab.add(tile_w3(st[p + "mlp.fc1.weight"])), "fc23": ab.add(tile_w3(st[p + "mlp.fc2.weight"])),
        })
    off["layers"] = layers
    convs = []
    for key, cin, cout, k, stride, tr in mimi_conv_specs():
        w, b = st[f"decoder.layers.{key}.conv.weight"], st[f"decoder.layers.{key}.conv.bias"]
        expect = (cin, cout, k) if tr else (cout, cin, k)
        if tuple(w.shape) != expect:
            raise ValueError(f"decoder.layers.{key}.conv.weight has shape {tuple(w.shape)}, expected {expect}")
        gw, gb = conv_as_gemm(w, b, tr, stride)
        convs.append({"w": ab.add(tile_t16x32(gw, f32)), "b": ab.add(gb), "cin": cin, "cout": cout, "k": k,
                      "stride": stride, "transposed": int(tr), "w3": ab.add(tile_w3(gw)) if gw.shape[0] >= 16 else 0})
    off["convs"] = convs
    # the output conv (64 -> 1, k3) once more as plain fp32 [3][64] (tap-major): the fused last SEANet stage applies it on the
    # vector units to the block's output while that is still in LDS
    fw, _ = conv_as_gemm(st["decoder.layers.14.conv.weight"], st["decoder.layers.14.conv.bias"], False, 1)
    off["final_w"] = ab.add(fw.reshape(-1).contiguous())
    off["n_layers"] = n_layers
    off["max_positions"] = max_positions
    if report is not None:
        report["unused"] = st.unused()
    return ab.finish(), off


ENC_RATIOS = (4, 5, 6, 8)


def mimi_encoder_conv_specs():
    """(HF key under encoder.layers, cin, cout, k, stride) in execution order after layer 0
    (codec/seanet.py:57-90): per ratio the resnet block's two convs and the strided conv, then the final conv."""
    specs, ch, li = [], 64, 1
    for r in ENC_RATIOS:
        specs.append((f"{li}.block.1", ch, ch // 2, 3, 1))
        specs.append((f"{li}.block.3", ch // 2, ch, 1, 1))
        specs.append((str(li + 2), ch, 2 * ch, 2 * r, r))
        ch *= 2
        li += 3
    specs.append(("14", 1024, 512, 3, 1))
    return specs


def pack_mimi_encoder(state: Dict[str, torch.Tensor], num_codebooks: int = 8, max_positions: int = 2048, report: Dict[str, object] = None):
    """Arena + offsets (SmolttsMimiEncWeights) of the encode half; ``state`` uses the Hugging Face
    ``MimiModel.state_dict()`` names (encoder.*, encoder_transformer.*, downsample.*, quantizer.*)."""
    st = _Reads({k: v.float() for k, v in state.items()})
    ab = ArenaBuilder()
    f32 = torch.float32
    off: Dict[str, object] = {}
    w0 = st["encoder.layers.0.conv.weight"]
    if tuple(w0.shape) != (64, 1, 7):
        raise ValueError(f"encoder.layers.0.conv.weight has shape {tuple(w0.shape)}, expected (64, 1, 7)")
    off["conv0_w"] = ab.add(torch.cat([w0[:, 0, :], torch.zeros(64, 1)], dim=1).contiguous())
    off["conv0_b"] = ab.add(st["encoder.layers.0.conv.bias"])
    convs = []
    for key, cin, cout, k, stride in mimi_encoder_conv_specs():
        w, b = st[f"encoder.layers.{key}.conv.weight"], st[f"encoder.layers.{key}.conv.bias"]
        if tuple(w.shape) != (cout, cin, k):
            raise ValueError(f"encoder.layers.{key}.conv.weight has shape {tuple(w.shape)}, expected {(cout, cin, k)}")
        gw, gb = conv_as_gemm(w, b, False, stride)
        convs.append({"w": ab.add(tile_t16x32(gw, f32)), "b": ab.add(gb), "cin": cin, "cout": cout, "k": k, "stride": stride,
                      "transposed": 0})
    off["convs"] = convs
    n_layers = 1 + max(int(k.split(".")[2]) for k in st if k.startswith("encoder_transformer.layers."))
    layers = []
    for li in range(n_layers):
        p = f"encoder_transformer.layers.{li}."
        wq, wk, wv = (st[p + f"self_attn.{n}_proj.weight"] for n in ("q", "k", "v"))
        layers.append({
            "ln1_w": ab.add(st[p + "input_layernorm.weight"]), "ln1_b": ab.add(st[p + "input_layernorm.bias"]),
            "wqkv": ab.add(tile_t16x32(torch.cat([_perm_heads(wq), _perm_heads(wk), wv]), f32)),
            "wo": ab.add(tile_t16x32(st[p + "self_attn.o_proj.weight"], f32)),
            "ls1": ab.add(st[p + "self_attn_layer_scale.scale"]),
            "ln2_w": ab.add(st[p + "post_attention_layernorm.weight"]), "ln2_b": ab.add(st[p + "post_attention_layernorm.bias"]),
            "fc1": ab.add(tile_t16x32(st[p + "mlp.fc1.weight"], f32)),
            "fc2": ab.add(tile_t16x32(st[p + "mlp.fc2.weight"], f32)),
            "ls2": ab.add(st[p + "mlp_layer_scale.scale"]),
        })
    off["layers"] = layers
    off["rope"] = ab.add(rope_table(max_positions, 64, 10000.0, bf16=False))
    wd = st["downsample.conv.weight"]  # (512, 512, 4), no bias
    off["downsample_w"] = ab.add(tile_t16x32(conv_as_gemm(wd, torch.zeros(wd.shape[0]), False, 2)[0], f32))
    off["in_proj"] = [ab.add(tile_t16x32(st[f"quantizer.{g}_residual_vector_quantizer.input_proj.weight"][:, :, 0].contiguous(), f32))
                      for g in ("semantic", "acoustic")]
    books = []
    for q in range(num_codebooks):
        grp, li = ("semantic", 0) if q == 0 else ("acoustic", q - 1)
        p = f"quantizer.{grp}_residual_vector_quantizer.layers.{li}.codebook."
        books.append(st[p + "embed_sum"] / torch.clamp(st[p + "cluster_usage"], min=1e-5)[:, None])  # rvq.py:41-48
    off["codebooks_t"] = ab.add(torch.stack([tile_t16x32(b, f32) for b in books]))
    off["codebooks"] = ab.add(torch.stack(books))
    off["codebook_sq"] = ab.add(torch.stack([(b * b).sum(-1) for b in books]))
    off["n_layers"] = n_layers
    off["max_positions"] = max_positions
    off["num_codebooks"] = num_codebooks
    if report is not None:
        report["unused"] = st.unused()
    return ab.finish(), off
