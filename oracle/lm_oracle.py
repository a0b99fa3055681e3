"""CPU oracle for the DualAR / RQ-Transformer greedy decode.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  ``smoltts_amd`` must never import it.

It is a plain PyTorch-eager fp32 restatement (KV-cached, batched) of the reference algorithm:

* arithmetic  : ``modeling/model/rq_transformer.py``  (RMSNorm :601-613, RoPE table :616-624 and
                rotation :627-640, Attention :535-570, FeedForward :573-582, DepthwiseLinear
                :585-598, embed :205-221, slow head / pre-norm hidden hand-off :249-260, fast path
                :411-448)
* decode loop : ``mlx_inference/src/smoltts_mlx/lm/generate.py:59-171`` (frame loop, 8 fast steps,
                depthwise fast-embedding offset :136-140, stop rule :162-166, frame budget :60,161),
                ``lm/rq_transformer.py:150-220`` (embed mask rule, depthwise head slice) and
                ``lm/cache.py:6-22`` (append-only KV cache; here a preallocated slab).

Pinning: ``tests/golden/make_lm_goldens.py`` (run in the build container, where /root/reference is
importable) checks this restatement against the *unmodified* reference ``RQTransformer.forward``
(teacher-forced self-consistency) and commits the resulting vectors under
``tests/golden/``.  The reference's own tests hold no vectors for this path.

Two reference quirks are switchable because the torch and MLX halves of the reference disagree:
``embed_mask`` ("torch": zero the codebook-embedding sum where code0 == 0, modeling :219; "mlx":
zero it where the text-row token is outside the semantic range, lm/rq_transformer.py:162-169) and
``rope_bf16`` (True: cos/sin table rounded to bf16 as modeling :624; False: exact fp32 as MLX
nn.RoPE).
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- config
@dataclass
class OracleLMConfig:
    """Field names and derived defaults follow modeling/model/rq_transformer.py:25-114."""

    vocab_size: int = 32000
    n_layer: int = 32
    n_head: int = 32
    dim: int = 4096
    intermediate_size: int = 16384
    n_local_heads: int = -1
    head_dim: int = 64
    rope_base: float = 10000
    norm_eps: float = 1e-5
    max_seq_len: int = 2048
    tie_word_embeddings: bool = True
    codebook_size: int = 160
    num_codebooks: int = 4
    fast_dim: Optional[int] = 1024
    n_fast_layer: int = 4
    fast_n_head: Optional[int] = 16
    fast_n_local_heads: Optional[int] = None
    fast_head_dim: Optional[int] = None
    fast_intermediate_size: Optional[int] = None
    depthwise_wte: Optional[bool] = False
    depthwise_output: Optional[bool] = False
    duplicate_code_0: Optional[bool] = True
    # token ids (byte-level tokenizer of data_pipeline/scripts/create_bytelevel_init.py:15-57)
    semantic_start_id: int = 320
    im_end_id: int = 270

    def __post_init__(self):
        if self.n_local_heads == -1:
            self.n_local_heads = self.n_head
        self.head_dim = self.dim // self.n_head  # modeling :65
        self.fast_dim = self.fast_dim or self.dim
        self.fast_n_head = self.fast_n_head or self.n_head
        self.fast_n_local_heads = self.fast_n_local_heads or self.n_local_heads
        self.fast_head_dim = self.fast_head_dim or self.head_dim
        self.fast_intermediate_size = self.fast_intermediate_size or self.intermediate_size
        if self.duplicate_code_0 is None:
            self.duplicate_code_0 = True

    @property
    def semantic_end_id(self) -> int:
        return self.semantic_start_id + self.codebook_size - 1

    @property
    def max_fast_seqlen(self) -> int:  # modeling :344-346
        return self.num_codebooks - (0 if self.duplicate_code_0 else 1)

    @classmethod
    def from_dict(cls, d: dict) -> "OracleLMConfig":
        known = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in known})

    @classmethod
    def from_json(cls, path) -> "OracleLMConfig":
        p = Path(path)
        if p.is_dir():
            p = p / "config.json"
        return cls.from_dict(json.loads(p.read_text()))


# ----------------------------------------------------------------------------- primitives
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """modeling :607-613: x.float() * rsqrt(mean(x^2) + eps), cast back, then * weight."""
    xf = x.float()
    out = (xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)).type_as(x)
    return out * weight


def rope_table(seq_len: int, n_elem: int, base: float, bf16: bool) -> Tensor:
    """modeling :616-624 -> (seq_len, n_elem/2, 2) [cos, sin]; fp32 values (bf16-rounded if bf16)."""
    freqs = 1.0 / (base ** (torch.arange(0, n_elem, 2)[: (n_elem // 2)].float() / n_elem))
    t = torch.arange(seq_len)
    freqs = torch.outer(t, freqs)
    fc = torch.polar(torch.ones_like(freqs), freqs)
    cache = torch.stack([fc.real, fc.imag], dim=-1)
    if bf16:
        cache = cache.to(torch.bfloat16)
    return cache.float()


def apply_rope(x: Tensor, cs: Tensor) -> Tensor:
    """modeling :627-640, interleaved pairs (2j, 2j+1).

    x: (..., H, hd); cs: broadcastable to (..., 1, hd/2, 2)."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    out = torch.stack(
        [
            xs[..., 0] * cs[..., 0] - xs[..., 1] * cs[..., 1],
            xs[..., 1] * cs[..., 0] + xs[..., 0] * cs[..., 1],
        ],
        -1,
    )
    return out.flatten(-2).type_as(x)


def _argmax_with_margin(logits: Tensor) -> Tuple[Tensor, Tensor]:
    """argmax (first maximal index, as torch.argmax) and the top-1/top-2 gap, per row."""
    top2 = torch.topk(logits, 2, dim=-1)
    ids = torch.argmax(logits, dim=-1)
    return ids, (top2.values[..., 0] - top2.values[..., 1])


# ----------------------------------------------------------------------------- model
class _Block:
    def __init__(self, st: Dict[str, Tensor], prefix: str, n_head: int, n_kv: int, hd: int, eps: float):
        g = lambda k: st[prefix + k].float()
        if prefix + "attention.wqkv.weight" in st:
            self.wqkv = g("attention.wqkv.weight")
        else:  # legacy split keys, modeling :528-533
            self.wqkv = torch.cat([g("attention.wq.weight"), g("attention.wk.weight"), g("attention.wv.weight")])
        self.wo = g("attention.wo.weight")
        self.w1 = g("feed_forward.w1.weight")
        self.w2 = g("feed_forward.w2.weight")
        self.w3 = g("feed_forward.w3.weight")
        self.attn_norm = g("attention_norm.weight")
        self.ffn_norm = g("ffn_norm.weight")
        self.n_head, self.n_kv, self.hd, self.eps = n_head, n_kv, hd, eps
        self.kv_bf16 = False  # engine option kv_dtype="bf16": K (after RoPE) and V are rounded to bf16 once, as they enter the cache

    def qkv(self, x: Tensor, cs: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """x (..., d) -> q (..., H, hd), k, v (..., KV, hd) with RoPE applied (modeling :542-552)."""
        qkv = rms_norm(x, self.attn_norm, self.eps) @ self.wqkv.T
        dq, dkv = self.n_head * self.hd, self.n_kv * self.hd
        q, k, v = qkv.split([dq, dkv, dkv], dim=-1)
        q = q.reshape(*q.shape[:-1], self.n_head, self.hd)
        k = k.reshape(*k.shape[:-1], self.n_kv, self.hd)
        v = v.reshape(*v.shape[:-1], self.n_kv, self.hd)
        q, k = apply_rope(q, cs), apply_rope(k, cs)
        if self.kv_bf16:  # round to nearest even, what the engine's cache write does; every use of K / V sees the rounded values
            k, v = k.bfloat16().float(), v.bfloat16().float()
        return q, k, v

    def mlp(self, h: Tensor) -> Tensor:
        hn = rms_norm(h, self.ffn_norm, self.eps)
        return h + (torch.nn.functional.silu(hn @ self.w1.T) * (hn @ self.w3.T)) @ self.w2.T


def _gqa_attend(q: Tensor, K: Tensor, V: Tensor, mask: Optional[Tensor]) -> Tensor:
    """q (B,S,H,hd); K,V (B,L,KV,hd); mask (B,1,S,L) bool True=keep. softmax(qk/sqrt(hd)) v.

    modeling :554-568 (repeat_interleave GQA, SDPA)."""
    B, S, H, hd = q.shape
    rep = H // K.shape[2]
    k = K.repeat_interleave(rep, dim=2).permute(0, 2, 1, 3)  # B,H,L,hd
    v = V.repeat_interleave(rep, dim=2).permute(0, 2, 1, 3)
    s = (q.permute(0, 2, 1, 3) @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B, S, H * hd)


@dataclass
class OracleFrameLog:
    """Per-utterance generation record."""

    grid: List[List[int]] = field(default_factory=list)  # one 9-column per frame
    min_margin: float = float("inf")
    done: bool = False

    def as_tensor(self) -> Tensor:
        return torch.tensor(self.grid, dtype=torch.int64).T.contiguous()  # (9, F)


class LMOracle:
    """Batched, KV-cached greedy DualAR decode in fp32 torch eager on CPU."""

    def __init__(
        self,
        cfg: OracleLMConfig,
        state: Dict[str, Tensor],
        embed_mask: str = "torch",
        rope_bf16: bool = True,
        kv_bf16: bool = False,
    ):
        """``kv_bf16``: mirror of the engine's bf16 KV-cache option (slow transformer only; the depth transformer's 8-entry
        per-frame cache stays fp32 there too).  The reference itself keeps K/V in the activation dtype (lm/cache.py:6-22)."""
        assert embed_mask in ("torch", "mlx")
        self.cfg, self.embed_mask = cfg, embed_mask
        st = {k.replace("_orig_mod.", ""): v for k, v in state.items()}
        f = lambda k: st[k].float()
        self.E_text = f("embeddings.weight")
        self.E_cb = f("codebook_embeddings.weight")
        self.E_fast = f("fast_embeddings.weight")
        self.norm_w = f("norm.weight")
        self.fast_norm_w = f("fast_norm.weight")
        self.out_w = self.E_text if cfg.tie_word_embeddings else f("output.weight")
        self.layers = [
            _Block(st, f"layers.{i}.", cfg.n_head, cfg.n_local_heads, cfg.head_dim, cfg.norm_eps)
            for i in range(cfg.n_layer)
        ]
        self.fast_layers = [
            _Block(st, f"fast_layers.{i}.", cfg.fast_n_head, cfg.fast_n_local_heads, cfg.fast_head_dim, cfg.norm_eps)
            for i in range(cfg.n_fast_layer)
        ]
        for L in self.layers:
            L.kv_bf16 = kv_bf16
        if "fast_project_in.weight" in st:  # modeling :339-342 (Linear *with* bias)
            self.proj_w, self.proj_b = f("fast_project_in.weight"), f("fast_project_in.bias")
        else:
            self.proj_w = self.proj_b = None
        fo = f("fast_output.weight")
        nfast, cs, fd = cfg.max_fast_seqlen, cfg.codebook_size, cfg.fast_dim
        if cfg.depthwise_output:
            if fo.dim() == 3:  # torch layout (n, d, 2048), einsum "ijm,jmk->ijk" modeling :598
                self.fast_out = fo.permute(0, 2, 1).contiguous()  # (n, 2048, d)
            else:  # MLX flattened (n*2048, d), train/convert_safetensors.py:10-15
                self.fast_out = fo.reshape(nfast, cs, fd)
        else:
            self.fast_out = fo.reshape(1, cs, fd).expand(nfast, cs, fd)
        self.rope = rope_table(cfg.max_seq_len, cfg.head_dim, cfg.rope_base, rope_bf16)
        self.fast_rope = rope_table(nfast, cfg.fast_head_dim, cfg.rope_base, rope_bf16)

    # ---- embed, modeling :205-221 / lm/rq_transformer.py:150-170
    def embed(self, cols: Tensor) -> Tensor:
        """cols (..., 9) int64 -> (..., d)."""
        cfg = self.cfg
        text = self.E_text[cols[..., 0]]
        off = torch.arange(0, cfg.num_codebooks * cfg.codebook_size, cfg.codebook_size)
        if not cfg.duplicate_code_0:
            off = off[1:]
        vq = self.E_cb[cols[..., 1:] + off].sum(dim=-2)
        if self.embed_mask == "torch":
            keep = cols[..., 1] != 0
        else:
            keep = (cols[..., 0] >= cfg.semantic_start_id) & (cols[..., 0] <= cfg.semantic_end_id)
        return text + vq * keep[..., None].to(vq.dtype)

    # ---- slow transformer
    def _alloc(self, B: int, cap: int):
        cfg = self.cfg
        shp = (B, cap, cfg.n_local_heads, cfg.head_dim)
        self.K = [torch.zeros(shp) for _ in range(cfg.n_layer)]
        self.V = [torch.zeros(shp) for _ in range(cfg.n_layer)]
        self.pos = torch.zeros(B, dtype=torch.int64)  # tokens cached so far, per utterance

    def prefill_one(self, b: int, grid: Tensor) -> Tensor:
        """grid (9, T) -> pre-norm hidden of the last prompt token (d,). Fills cache rows [0,T) of b."""
        T = grid.shape[1]
        x = self.embed(grid.T.contiguous())[None]  # 1,T,d
        cs = self.rope[:T][None, :, None]  # 1,T,1,hd/2,2
        mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None]
        for li, L in enumerate(self.layers):
            q, k, v = L.qkv(x, cs)
            self.K[li][b, :T], self.V[li][b, :T] = k[0], v[0]
            a = _gqa_attend(q, k, v, mask)
            x = L.mlp(x + a @ L.wo.T)
        self.pos[b] = T
        return x[0, -1]

    def decode_cols(self, cols: Tensor) -> Tensor:
        """cols (B, 9): one new token per utterance at position pos[b] -> pre-norm hidden (B, d)."""
        B = cols.shape[0]
        x = self.embed(cols)[:, None]  # B,1,d
        cs = self.rope[self.pos][:, None, None]  # B,1,1,hd/2,2
        Lmax = int(self.pos.max()) + 1
        mask = (torch.arange(Lmax)[None] <= self.pos[:, None])[:, None, None]  # B,1,1,Lmax
        bi = torch.arange(B)
        for li, L in enumerate(self.layers):
            q, k, v = L.qkv(x, cs)
            self.K[li][bi, self.pos], self.V[li][bi, self.pos] = k[:, 0], v[:, 0]
            a = _gqa_attend(q, self.K[li][:, :Lmax], self.V[li][:, :Lmax], mask)
            x = L.mlp(x + a @ L.wo.T)
        self.pos += 1
        return x[:, 0]

    def slow_head(self, hidden: Tensor) -> Tensor:
        """modeling :249-255 / lm/rq_transformer.py:184-189: logits = RMSNorm(x) E^T."""
        return rms_norm(hidden, self.norm_w, self.cfg.norm_eps) @ self.out_w.T

    # ---- fast (depth) transformer, lm/generate.py:110-141 + lm/rq_transformer.py:194-220
    def fast_decode(self, hidden: Tensor) -> Tuple[Tensor, Tensor]:
        """hidden (B, d) pre-norm slow hidden -> codes (B, n_fast) int64, min margins (B,)."""
        cfg = self.cfg
        B, n = hidden.shape[0], cfg.max_fast_seqlen
        x = hidden if self.proj_w is None else hidden @ self.proj_w.T + self.proj_b
        Kc = [torch.zeros(B, n, cfg.fast_n_local_heads, cfg.fast_head_dim) for _ in self.fast_layers]
        Vc = [torch.zeros_like(k) for k in Kc]
        codes = torch.zeros(B, n, dtype=torch.int64)
        margin = torch.full((B,), float("inf"))
        for i in range(n):
            h = x[:, None]
            cs = self.fast_rope[i][None, None, None]
            for li, L in enumerate(self.fast_layers):
                q, k, v = L.qkv(h, cs)
                Kc[li][:, i], Vc[li][:, i] = k[:, 0], v[:, 0]
                a = _gqa_attend(q, Kc[li][:, : i + 1], Vc[li][:, : i + 1], None)
                h = L.mlp(h + a @ L.wo.T)
            logits = rms_norm(h[:, 0], self.fast_norm_w, cfg.norm_eps) @ self.fast_out[i].T
            ids, m = _argmax_with_margin(logits)
            codes[:, i] = ids
            margin = torch.minimum(margin, m)
            if i + 1 < n:
                if cfg.depthwise_wte:  # lm/generate.py:136-140
                    off = i if cfg.duplicate_code_0 else i + 1
                    x = self.E_fast[ids + max(0, off * cfg.codebook_size)]
                else:
                    x = self.E_fast[ids]
        return codes, margin

    # ---- frame loop, lm/generate.py:59-171
    @torch.no_grad()
    def generate(
        self,
        prompts: Sequence[Tensor],
        max_frames: int,
        stop_on_eos: bool = True,
        cache_cap: Optional[int] = None,
    ) -> List[OracleFrameLog]:
        """prompts: list of (9, T_b) int grids. Runs at most ``max_frames`` frames per utterance.

        An utterance stops *after* emitting the frame whose slow id is <|im_end|> (generate.py:162-166)
        when ``stop_on_eos``; finished utterances keep their slot (their later columns are ignored)."""
        B = len(prompts)
        cap = cache_cap or (max(int(p.shape[1]) for p in prompts) + max_frames + 1)
        self._alloc(B, cap)
        logs = [OracleFrameLog() for _ in range(B)]
        hidden = torch.stack([self.prefill_one(b, prompts[b].long()) for b in range(B)])
        for f in range(max_frames):
            ids, m_slow = _argmax_with_margin(self.slow_head(hidden))
            codes, m_fast = self.fast_decode(hidden)
            cols = torch.cat([ids[:, None], codes], dim=1)  # B, 1+n_fast
            for b in range(B):
                if logs[b].done:
                    continue
                logs[b].grid.append(cols[b].tolist())
                logs[b].min_margin = min(logs[b].min_margin, float(m_slow[b]), float(m_fast[b]))
                if stop_on_eos and int(ids[b]) == self.cfg.im_end_id:
                    logs[b].done = True
            if all(l.done for l in logs) or f + 1 == max_frames:
                break
            hidden = self.decode_cols(cols)
        return logs

    # ---- teacher-forced logits (for cross-checks against the reference's RQTransformer.forward)
    @torch.no_grad()
    def teacher_forced(self, grid: Tensor) -> Tuple[Tensor, Tensor]:
        """grid (9, S) -> token_logits (S, V), codebook_logits (S, n_fast, 2048).

        Position s's fast pass sees hidden[s] followed by the fast embeddings of codes at s+1
        (modeling :417-422); the last position sees zeros-padded codes."""
        cfg = self.cfg
        self._alloc(1, grid.shape[1])
        S = grid.shape[1]
        x = self.embed(grid.T.contiguous().long())[None]
        cs = self.rope[:S][None, :, None]
        mask = torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
        for L in self.layers:
            q, k, v = L.qkv(x, cs)
            x = L.mlp(x + _gqa_attend(q, k, v, mask) @ L.wo.T)
        tok = self.slow_head(x[0])
        n = cfg.max_fast_seqlen
        nxt = torch.zeros(S, grid.shape[0] - 2, dtype=torch.int64)
        nxt[:-1] = grid[1:-1, 1:].T.long()
        off = torch.arange(0, cfg.codebook_size * (cfg.num_codebooks - 1), cfg.codebook_size)
        if not cfg.duplicate_code_0:
            off = off[1:]
        fe = self.E_fast[nxt + off]  # S, n-1, d   (modeling :418-421)
        fe[-1] = self.E_fast[torch.zeros(fe.shape[1], dtype=torch.int64)]  # F.pad(..., value=0) column
        hid = x[0] if self.proj_w is None else x[0] @ self.proj_w.T + self.proj_b
        h = torch.cat([hid[:, None], fe], dim=1)  # S, n, d
        csf = self.fast_rope[:n][None, :, None]
        fmask = torch.tril(torch.ones(n, n, dtype=torch.bool))[None, None]
        for L in self.fast_layers:
            q, k, v = L.qkv(h, csf)
            h = L.mlp(h + _gqa_attend(q, k, v, fmask) @ L.wo.T)
        hn = rms_norm(h, self.fast_norm_w, cfg.norm_eps)
        cb = torch.einsum("snd,nkd->snk", hn, self.fast_out)
        # the zero-padded last column is dropped before the fast layers and scattered back as
        # zeros (modeling :426-436, :451-467)
        cb[-1] = 0
        return tok, cb
