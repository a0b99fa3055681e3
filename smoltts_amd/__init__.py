"""smoltts_amd: MI355X-native DualAR + Mimi decode hot path.

``SmolTTS`` mirrors the reference façade mlx_inference/src/smoltts_mlx/__init__.py:25-151: same
constructor arguments, ``__call__(input, voice) -> float32 PCM`` and ``stream(input, voice)`` yielding
80 ms chunks (1920 samples at 24 kHz).  Everything heavy runs in ``libsmoltts_hip.so``.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, List, Optional

__version__ = "0.1.0"


class SmolTTS:
    def __init__(self, model_id: Optional[str] = None, checkpoint_dir: Optional[str] = None,
                 mimi_checkpoint: Optional[str] = None, numerics=None, state=None, config=None, mimi_state=None,
                 codec_window: int = 0, weight_format: str = "bf16", verbose: bool = False):
        """``checkpoint_dir``: config.json + tokenizer.json + model.safetensors | model.pth (reference
        layouts).  ``mimi_checkpoint``: the Hugging Face kyutai/mimi ``model.safetensors`` (or its
        directory).  ``state``/``config``/``mimi_state`` allow in-memory (e.g. synthetic) weights.
        ``model_id`` (Hugging Face download in the reference) needs network access and is refused.
        ``weight_format="fp8"`` stores the Linears as e4m3 with per-row scales (``packing.fp8_reference_state`` is the model
        then computed)."""
        import torch  # noqa: F401

        from .checkpoint import load_checkpoint, load_mimi_state
        from .config import NumericsMode, TokenConfig
        from .engine import LMEngine, MimiEngine
        from .prompt import PromptEncoder
        from .tokenizer import load_tokenizer

        if checkpoint_dir is None and state is None:
            raise ValueError("pass checkpoint_dir (or state+config); downloading model_id=%r needs network access" % (model_id,))
        if checkpoint_dir is not None:
            config, tokenizer, state = load_checkpoint(checkpoint_dir)
        else:
            tokenizer = load_tokenizer(None, config.codebook_size)
        if mimi_state is None:
            if mimi_checkpoint is None:
                raise ValueError("pass mimi_checkpoint (kyutai/mimi model.safetensors) or mimi_state")
            mimi_state = load_mimi_state(mimi_checkpoint)
        self.config = config
        self.tokenizer = tokenizer
        self.token_config = TokenConfig.from_tokenizer(tokenizer, config)
        self.lm = LMEngine(config, state, self.token_config, numerics or NumericsMode.torch_reference(), weight_format=weight_format)
        self.prompt_encoder = PromptEncoder.from_config(tokenizer, config, self.token_config)
        self.codec = MimiEngine(mimi_state, num_codebooks=config.num_codebooks, window=codec_window, max_positions=2 * (1026 + 64))  # 1025 frames + a scheduler tick of overshoot, 2 positions per frame
        # the encode half (voice-clone prompts) is packed on first use, and only if the checkpoint carries it
        self._mimi_encoder_state = mimi_state if "encoder.layers.0.conv.weight" in mimi_state else None
        self._codec_window = codec_window
        self._encoder = None
        self.sampling_rate = 24_000
        self.verbose = verbose  # print the reference's per-call timing lines (lm/generate.py:187-214)
        self.last_stats: dict = {}  # timing of the last generate_codes / __call__ (BatchGenerator.stats + codec_ms)

    # -- prompt (``_get_prompt``, __init__.py:120-151)
    def _get_prompt(self, input: str, voice: str, sysprompt=None):
        return self.prompt_encoder.build_prompt(input, voice, sysprompt)

    def _settings(self, generation_settings):
        from .config import GenerationSettings

        # the reference façade always uses GenerationSettings() (temp 0.7 / 0.7, __init__.py:77,85)
        return generation_settings or GenerationSettings()

    def generate_codes(self, inputs: List[str], voices: Optional[List[str]] = None, generation_settings=None, speakers=None):
        """Batched synthesis to audio-code grids: one (n_codebooks, F_b) uint32 array per input."""
        import numpy as np

        from .generate import BatchGenerator

        voices = voices or ["heart"] * len(inputs)
        speakers = speakers or [None] * len(inputs)
        prompts = [self._get_prompt(t, v, sp) for t, v, sp in zip(inputs, voices, speakers)]
        gen = BatchGenerator(self.lm, prompts, self._settings(generation_settings), frames_per_sync=16)
        cols: List[list] = [[] for _ in inputs]
        for row in gen:
            for b, tok in enumerate(row):
                if tok is not None and tok.audio_codes is not None:
                    cols[b].append(tok.audio_codes[0, :, 0])
        self.last_stats = gen.stats()
        gen.close()
        if self.verbose and self.last_stats:
            st = self.last_stats
            print(f"Prompt: {st['prompt_tokens']} tokens in {st['prefill_ms']:.1f} ms ({st['prefill_tokens_per_s']:.0f} tokens/s)")
            print(f"Generated {st['frames']} frames: {st['frames_per_s']:.1f} frames/s, {st['ms_per_frame_step']:.3f} ms/frame-step, "
                  f"{st['realtime_x']:.1f}x realtime")
        nq = self.config.num_codebooks
        return [np.stack(c, axis=1).astype(np.uint32) if c else np.zeros((nq, 0), np.uint32) for c in cols]

    def decode_codes(self, codes) -> "np.ndarray":
        """(n_codebooks, F) -> float32 PCM (1920 F,)  == codec.decode(gen) of the reference."""
        import numpy as np
        import torch

        from .engine import MimiSession

        F_ = int(codes.shape[1])
        if F_ == 0:
            return np.zeros(0, np.float32)
        import time

        t0 = time.perf_counter()
        sess = MimiSession(self.codec, max_batch=1, max_chunk_frames=min(16, F_))
        dev = torch.from_numpy(np.ascontiguousarray(codes.T.astype(np.int32)))[None].cuda()
        pcm = sess.decode(dev).cpu().numpy().reshape(-1)
        sess.close()
        self.last_stats = dict(self.last_stats, codec_ms=(time.perf_counter() - t0) * 1e3, codec_frames=F_)
        if self.verbose:
            print(f"Decoded {F_} frames to PCM in {self.last_stats['codec_ms']:.1f} ms")
        return pcm

    def __call__(self, input: str, voice: Optional[str] = "heart", speaker=None, generation_settings=None):
        """Returns flattened float32 PCM (reference __call__, __init__.py:64-81)."""
        codes = self.generate_codes([input], [voice if voice is not None else "heart"], generation_settings,
                                    speakers=None if speaker is None else [speaker])[0]
        return self.decode_codes(codes)

    # -- voice-clone prompts (``create_speaker``, __init__.py:97-118)
    def encode_audio(self, audio) -> "np.ndarray":
        """24 kHz mono float PCM (any shape, flattened) -> (n_codebooks, F) uint32 Mimi codes (``codec.encode``)."""
        import numpy as np

        from .engine import MimiEncoder

        if self._encoder is None:
            if self._mimi_encoder_state is None:
                raise ValueError("the Mimi checkpoint has no encoder.* weights: voice-clone prompts need the full kyutai/mimi model")
            self._encoder = MimiEncoder(self._mimi_encoder_state, num_codebooks=8, window=self._codec_window)
            self._mimi_encoder_state = None
        pcm = np.asarray(audio, dtype=np.float32).reshape(-1)
        return self._encoder.encode(pcm).cpu().numpy().astype(np.uint32)

    def create_speaker(self, samples: List[dict], system_prompt: Optional[str] = None) -> "np.ndarray":
        """Speaker prompt grid from reference recordings: per sample the user turn with its transcript, then
        its Mimi codes closed by ``<|im_end|>\\n``; optionally a leading system turn.  Pass the result as
        ``speaker=`` to ``__call__``."""
        import numpy as np

        turns = []
        for sample in samples:
            if "audio" not in sample or "text" not in sample:
                raise ValueError(f"Sample must contain both 'text' and 'audio' but got {sample.keys()}")
            turns.append(self.prompt_encoder.encode_text_turn("user", sample["text"]))
            turns.append(self.prompt_encoder.encode_vq(self.encode_audio(sample["audio"])[:8, :].astype(np.int64)))
        if system_prompt is not None:
            turns = [self.prompt_encoder.encode_text_turn("system", system_prompt), *turns]
        return np.concatenate(turns, axis=1).astype(np.int32)

    def stream(self, input: str, voice: Optional[str] = "heart", generation_settings=None, overlap: bool = True,
               reference_upsample: bool = False) -> Iterator["np.ndarray"]:
        """Yields one 1920-sample float32 chunk per generated frame, including the terminating
        <|im_end|> frame (reference stream, __init__.py:83-95, decodes vq_tensor[:, 1:, :] of every
        frame).  The codec carries its streaming state, so the chunks concatenate to the batch decode.
        ``reference_upsample``: up-sample every frame on its own as the reference's ``decode_step`` does (codec/mimi.py:77:
        no tap overlap from the previous frame) -- the chunks then equal the reference's own ``stream`` and no longer its batch
        decode; a quirk kept switchable like ``NumericsMode``'s.
        ``overlap``: the codec step of frame f runs beside frame f + 1 on a second stream (``generate.stream_pcm``); the chunks
        are the same numbers either way."""
        import numpy as np

        from .engine import LMSession, MimiSession
        from .generate import _apply_sampling, stream_pcm

        prompt = np.asarray(self._get_prompt(input, voice if voice is not None else "0"))
        if prompt.ndim == 3:
            prompt = prompt[0]
        settings = self._settings(generation_settings)
        max_new = settings.max_new_tokens if settings.max_new_tokens is not None else self.config.max_seq_len
        T = int(prompt.shape[1])
        sess = LMSession(self.lm, 1, max_seq=min(self.config.max_seq_len, T + max_new + 2), max_rows=T, max_frames=max_new + 1)
        _apply_sampling(sess, settings)
        msess = MimiSession(self.codec, max_batch=1, max_chunk_frames=1, stateless_upsample=reference_upsample)
        try:
            yield from stream_pcm(sess, msess, prompt, stop_on_eos=True, overlap=overlap)
        finally:
            msess.close()
            sess.close()
