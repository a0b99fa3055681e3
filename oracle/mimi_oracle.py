"""CPU oracle for the Mimi codec *decode* half.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Restates, in plain fp32 PyTorch on CPU, the reference chain ``MimiModel._decode_frame``
(mlx_inference/src/smoltts_mlx/codec/mimi.py:73-104):

* RVQ decode       codec/rvq.py:41-52 (embed = embed_sum / max(cluster_usage, 1e-5)), :118-131
                   (sum of codebook rows, then 1x1 ``output_proj``), :179-186 (semantic codebook 0 +
                   acoustic codebooks 1..7)
* upsample         codec/conv.py:232-282 (depthwise ConvTranspose1d k=4 s=2, trim k-s on the right)
* transformer      codec/transformer.py:34-150 (pre-LayerNorm, half-split RoPE base 1e4, 8 heads x 64,
                   erf-GELU MLP, per-channel layer scale); ``window`` = 0 reproduces the MLX code (the
                   declared context=250 is never applied, :22), ``window`` = 250 reproduces
                   ``transformers.MimiModel`` (sliding_window)
* SEANet decoder   codec/seanet.py:99-161, codec/conv.py:68-220 (causal left padding k_eff - stride,
                   ConvTranspose trim k - s on the right, ELU, residual blocks)

The arithmetic itself lives in a third-party dependency of the reference (``transformers``,
pinned 4.48.3 in the reference's uv.lock; weights ``kyutai/mimi``, unavailable offline).  The
restatement is pinned by ``tests/test_mimi_oracle.py`` against ``transformers.MimiModel`` (as
installed, 5.x) on seeded random weights for F <= 125 frames, and by committed goldens
(``tests/golden/mimi_*.npz``).  The reference's own tests hold no vectors for this path.

Weights are a dict with the Hugging Face ``MimiModel.state_dict()`` key names (the contract
``load_mimi`` consumes, codec/mimi.py:107-156), in torch layouts.

Streaming: the reference's ``decode_step`` up-samples each frame statelessly (mimi.py:77), so its
streaming output differs from its own batch decode.  This oracle defines streaming as the causal
prefix property instead: chunk f of a stream == samples [1920 f, 1920 (f+1)) of the batch decode.
``decode(codes, upsample_call_frames=c)`` restates the reference's own streaming instead: the up-sampling
sees c frames at a time (c = 1: ``SmolTTS.stream``, __init__.py:83-95), everything behind it is causal
with carried state (transformer cache, ``decoder.step``) and therefore equals the batch computation over
the concatenated rows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

RATIOS = (8, 6, 5, 4)
N_FILTERS = 64
DIMENSION = 512
SAMPLES_PER_FRAME = 1920


def _causal_conv1d(x: Tensor, w: Tensor, b: Optional[Tensor], dilation: int = 1) -> Tensor:
    """MimiConv1d.__call__ for stride 1 (conv.py:120-128): all padding k_eff-1 on the left."""
    k_eff = (w.shape[-1] - 1) * dilation + 1
    return F.conv1d(F.pad(x, (k_eff - 1, 0)), w, b, dilation=dilation)


def _causal_convtr1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, groups: int = 1) -> Tensor:
    """MimiConvTranspose1d.__call__ (conv.py:195-199): full transposed conv, drop k-s on the right."""
    y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
    k = w.shape[-1]
    return y[..., : y.shape[-1] - (k - stride)]


class MimiDecodeOracle:
    def __init__(self, state: Dict[str, Tensor], num_codebooks: int = 8, window: int = 0):
        self.st = {k: v.float() for k, v in state.items()}
        self.nq = num_codebooks
        self.window = window
        self.n_layers = 1 + max(
            int(k.split(".")[2]) for k in self.st if k.startswith("decoder_transformer.layers.")
        )

    # -- rvq.py:41-52,118-131,179-186
    def _codebook(self, prefix: str) -> Tensor:
        es, cu = self.st[prefix + "embed_sum"], self.st[prefix + "cluster_usage"]
        return es / torch.clamp(cu, min=1e-5)[:, None]

    def rvq_decode(self, codes: Tensor) -> Tensor:
        """codes (B, nq, F) -> (B, 512, F)."""
        B, nq, Fr = codes.shape
        sem = "quantizer.semantic_residual_vector_quantizer."
        aco = "quantizer.acoustic_residual_vector_quantizer."
        q_sem = self._codebook(sem + "layers.0.codebook.")[codes[:, 0]]  # B,F,256
        q_aco = torch.zeros_like(q_sem)
        for i in range(1, nq):
            q_aco = q_aco + self._codebook(aco + f"layers.{i - 1}.codebook.")[codes[:, i]]
        out = F.conv1d(q_sem.transpose(1, 2), self.st[sem + "output_proj.weight"])
        out = out + F.conv1d(q_aco.transpose(1, 2), self.st[aco + "output_proj.weight"])
        return out

    # -- conv.py:232-282
    def upsample(self, x: Tensor) -> Tensor:
        w = self.st["upsample.conv.weight"]  # (512, 1, 4)
        return _causal_convtr1d(x, w, None, stride=2, groups=w.shape[0])

    # -- transformer.py
    def _rope_half(self, x: Tensor, pos: Tensor) -> Tensor:
        """nn.RoPE(traditional=False): pairs (j, j+hd/2); x (B,H,T,hd), pos (T,)."""
        hd = x.shape[-1]
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
        ang = pos.float()[:, None] * inv[None]  # T, hd/2
        c, s = torch.cos(ang), torch.sin(ang)
        x1, x2 = x[..., : hd // 2], x[..., hd // 2 :]
        return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)

    def transformer(self, x: Tensor) -> Tensor:
        """x (B, T, 512) -> (B, T, 512); causal (optionally windowed) self-attention."""
        B, T, D = x.shape
        H, hd = 8, 64
        pos = torch.arange(T)
        i, j = pos[:, None], pos[None, :]
        keep = j <= i
        if self.window:
            keep = keep & (j > i - self.window)
        for li in range(self.n_layers):
            p = f"decoder_transformer.layers.{li}."
            g = lambda k: self.st[p + k]
            h = F.layer_norm(x, (D,), g("input_layernorm.weight"), g("input_layernorm.bias"), 1e-5)
            q = (h @ g("self_attn.q_proj.weight").T).view(B, T, H, hd).transpose(1, 2)
            k = (h @ g("self_attn.k_proj.weight").T).view(B, T, H, hd).transpose(1, 2)
            v = (h @ g("self_attn.v_proj.weight").T).view(B, T, H, hd).transpose(1, 2)
            q, k = self._rope_half(q, pos), self._rope_half(k, pos)
            s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
            s = s.masked_fill(~keep, float("-inf"))
            a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
            x = x + g("self_attn_layer_scale.scale") * (a @ g("self_attn.o_proj.weight").T)
            h = F.layer_norm(x, (D,), g("post_attention_layernorm.weight"), g("post_attention_layernorm.bias"), 1e-5)
            m = F.gelu(h @ g("mlp.fc1.weight").T) @ g("mlp.fc2.weight").T
            x = x + g("mlp_layer_scale.scale") * m
        return x

    # -- seanet.py:99-161
    def seanet(self, x: Tensor) -> Tensor:
        """x (B, 512, T) -> (B, 1, 960 T)."""
        g = lambda k: self.st["decoder.layers." + k]
        x = _causal_conv1d(x, g("0.conv.weight"), g("0.conv.bias"))
        li = 1
        for r in RATIOS:
            x = F.elu(x)  # layer li
            x = _causal_convtr1d(x, g(f"{li + 1}.conv.weight"), g(f"{li + 1}.conv.bias"), stride=r)
            res = x  # resnet block li+2: ELU, conv k3, ELU, conv k1, + residual
            y = _causal_conv1d(F.elu(x), g(f"{li + 2}.block.1.conv.weight"), g(f"{li + 2}.block.1.conv.bias"))
            y = _causal_conv1d(F.elu(y), g(f"{li + 2}.block.3.conv.weight"), g(f"{li + 2}.block.3.conv.bias"))
            x = res + y
            li += 3
        x = F.elu(x)  # layer 13
        return _causal_conv1d(x, g("14.conv.weight"), g("14.conv.bias"))

    # -- mimi.py:73-104
    @torch.no_grad()
    def decode(self, codes: Tensor, upsample_call_frames: Optional[int] = None) -> Tensor:
        """codes (B, nq, F) int -> pcm (B, 1, 1920 F) float32.  ``upsample_call_frames`` = c: the PCM a stream of
        ``decode_step`` calls of c frames each yields in the reference (mimi.py:73-77: ``self.upsample`` on the call's frames
        alone, i.e. taps 2, 3 of a call's last frame never reach the next call's first two rows)."""
        e = self.rvq_decode(codes.long())
        if upsample_call_frames:
            c = int(upsample_call_frames)
            e = torch.cat([self.upsample(e[:, :, i:i + c]) for i in range(0, e.shape[2], c)], dim=2)
        else:
            e = self.upsample(e)
        e = self.transformer(e.transpose(1, 2)).transpose(1, 2)
        return self.seanet(e)

    @torch.no_grad()
    def intermediates(self, codes: Tensor) -> Dict[str, Tensor]:
        """Stage outputs for kernel-level parity tests (channel-last, (B, T, C))."""
        e0 = self.rvq_decode(codes.long())
        e1 = self.upsample(e0)
        e2 = self.transformer(e1.transpose(1, 2))
        pcm = self.seanet(e2.transpose(1, 2))
        return {"rvq": e0.transpose(1, 2), "upsample": e1.transpose(1, 2), "transformer": e2, "pcm": pcm}


# =============================================================================================
# Encode half (voice-clone prompts): MimiModel.encode, mlx_inference/src/smoltts_mlx/codec/mimi.py:64-71
# =============================================================================================
def _extra_padding(length: int, k_eff: int, stride: int) -> int:
    """MimiConv1d._get_extra_padding_for_conv1d (conv.py:112-118)."""
    padding_total = k_eff - stride
    n_frames = math.ceil((length - k_eff + padding_total) / stride + 1) - 1
    return n_frames * stride + k_eff - padding_total - length


def _pad_conv1d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1, mode: str = "constant", extra_right: bool = False) -> Tensor:
    """MimiConv1d.__call__ (conv.py:120-128) through causal_pad1d (conv.py:25-41), which puts ALL padding —
    k_eff - stride plus the stride-alignment extra — on the left (``pad_tuple[-2] = (left + right, 0)``).
    ``extra_right`` puts the extra on the right instead, as ``transformers.MimiConv1d`` does; the two agree
    whenever every strided conv sees a length that is a multiple of its stride (audio length % 1920 == 0)."""
    k = w.shape[-1]
    extra = _extra_padding(x.shape[-1], k, stride)
    left, right = (k - stride, extra) if extra_right else (k - stride + extra, 0)
    if mode == "constant":
        x = F.pad(x, (left, right))
    else:  # "edge" (mimi.py:45): repeat the first / last sample
        x = F.pad(x, (left, right), mode="replicate")
    return F.conv1d(x, w, b, stride=stride)


class MimiEncodeOracle:
    """pcm (B, 1, L) -> codes (B, nq, ceil(L / 1920)).  TEST INFRASTRUCTURE ONLY (see the module header).

    * SEANet encoder   codec/seanet.py:52-96 (conv k7; per ratio 4,5,6,8: resnet block, ELU, strided conv
                       k = 2*ratio; ELU; conv k3), padding as ``_pad_conv1d``
    * transformer      codec/transformer.py (same block as the decoder's; causal; ``window`` as above)
    * downsample       codec/mimi.py:37-46 (Conv1d 512->512, k4, stride 2, no bias, "edge" padding)
    * RVQ encode       codec/rvq.py:16-22 (cdist), :54-64 (argmin), :99-116 (residual loop after the
                       group's 1x1 ``input_proj``), :157-177 (semantic group on the embeddings, acoustic
                       group on the same embeddings, not on the semantic residual)
    Pinned by tests/golden/mimi_enc_hf.npz (``transformers.MimiModel.encode`` on seeded weights).
    """

    def __init__(self, state: Dict[str, Tensor], num_codebooks: int = 8, window: int = 0, extra_right: bool = False):
        self.st = {k: v.float() for k, v in state.items()}
        self.nq = num_codebooks
        self.extra_right = extra_right
        # the encoder transformer is the decoder's block with other weights: reuse that restatement
        self._tr = MimiDecodeOracle({k.replace("encoder_transformer.", "decoder_transformer."): v for k, v in self.st.items()
                                     if k.startswith("encoder_transformer.")}, num_codebooks, window)

    def seanet(self, x: Tensor) -> Tensor:
        """x (B, 1, L) -> (B, 512, ceil(L/960))."""
        g = lambda k: self.st["encoder.layers." + k]
        pc = lambda x, key, stride=1: _pad_conv1d(x, g(key + ".conv.weight"), g(key + ".conv.bias"), stride, extra_right=self.extra_right)
        x = pc(x, "0")
        li = 1
        for r in reversed(RATIOS):
            y = pc(F.elu(x), f"{li}.block.1")
            y = pc(F.elu(y), f"{li}.block.3")
            x = x + y
            x = pc(F.elu(x), str(li + 2), r)
            li += 3
        return pc(F.elu(x), "14")

    def downsample(self, x: Tensor) -> Tensor:
        return _pad_conv1d(x, self.st["downsample.conv.weight"], None, 2, mode="edge", extra_right=self.extra_right)

    def embeddings(self, pcm: Tensor) -> Tensor:
        """Pre-quantisation latents (B, 512, F)."""
        e = self.seanet(pcm.float())
        e = self._tr.transformer(e.transpose(1, 2)).transpose(1, 2)
        return self.downsample(e)

    def _codebook(self, prefix: str) -> Tensor:
        es, cu = self.st[prefix + "embed_sum"], self.st[prefix + "cluster_usage"]
        return es / torch.clamp(cu, min=1e-5)[:, None]

    def rvq_encode(self, emb: Tensor, return_margin: bool = False):
        """emb (B, 512, F) -> codes (B, nq, F) [, smallest best-vs-second distance gap seen]."""
        B, _, Fr = emb.shape
        out, min_gap = [], float("inf")
        for grp, n in (("semantic", 1), ("acoustic", self.nq - 1)):
            p = f"quantizer.{grp}_residual_vector_quantizer."
            r = F.conv1d(emb, self.st[p + "input_proj.weight"]).transpose(1, 2).reshape(B * Fr, -1)  # rows (b, f)
            for i in range(n):
                cb = self._codebook(p + f"layers.{i}.codebook.")
                d2 = (r * r).sum(-1, keepdim=True) + (cb * cb).sum(-1)[None, :] - 2.0 * (r @ cb.T)
                d = d2.sqrt()  # rvq.py:22
                idx = d.argmin(-1)
                if return_margin:
                    top2 = torch.topk(d2, 2, dim=-1, largest=False).values
                    min_gap = min(min_gap, float((top2[:, 1] - top2[:, 0]).min()))
                r = r - cb[idx]
                out.append(idx.view(B, Fr))
        codes = torch.stack(out, dim=1)
        return (codes, min_gap) if return_margin else codes

    @torch.no_grad()
    def encode(self, pcm: Tensor) -> Tensor:
        return self.rvq_encode(self.embeddings(pcm))
