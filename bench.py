#!/usr/bin/env python3
"""Throughput bench of the hot path: greedy DualAR decode + Mimi decode to PCM.

Metric (BASELINE.json): Mimi frames/s (= 12.5 x real-time factor), whole job over all GPUs.
Workload at N=1 (BASELINE.json configs[2]): smoltts_byte_150m, bf16 weights, B=32 concurrent
utterances, synthetic ChatML prompts, seeded random weights, greedy, EOS disabled.

One *step* = one chunk of CH frames for all B slots of the rank: CH replays of the captured
frame graph (slow step + 8 depth steps + on-device argmax each) followed by one Mimi chunk decode
of those CH x B frames to PCM, everything resident in HBM.  Prompt prefill happens before the
timed region (the reference's own "x realtime" excludes it, lm/generate.py:199-214) and is
reported separately.  N > 1: one process per GPU, utterances sharded by rank (weak scaling, no
per-step collective), weights broadcast from rank 0 over RCCL.

Output: ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel: the fused
RMSNorm + w1|w3 GEMM + SwiGLU, timed in situ with HIP events) and `cpu_baseline` (the CPU oracle,
fp32 torch eager, timed on this host on a bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads():
    """CPU threads this process may really use (cgroup/affinity aware, capped at the 1-GPU share of 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def make_prompts(pe, n_total, seed=2):
    """voice u mod 11, printable-ASCII text of length U{40..160}."""
    from smoltts_amd.prompt import VOICES

    rng = np.random.default_rng(seed)
    out = []
    for u in range(n_total):
        n = int(rng.integers(40, 161))
        text = "".join(chr(int(c)) for c in rng.integers(32, 127, size=n))
        out.append(pe.build_prompt(text, VOICES[u % len(VOICES)]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="smoltts_byte_150m")
    ap.add_argument("--batch", type=int, default=32, help="utterance slots per GPU")
    ap.add_argument("--chunk", type=int, default=32, help="frames per step")
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the first-audio-chunk latency measurement")
    ap.add_argument("--no-mimi", action="store_true", help="diagnostic only: skip the Mimi decode (the result line is then not the metric)")
    ap.add_argument("--overlap-mimi", action="store_true", help="run the Mimi chunk decode on its own stream behind an event (measured: no gain, the many-workgroup Mimi kernels delay the latency-bound frame graphs)")
    ap.add_argument("--overlap-wait", default="device", choices=["device", "host"], help="with --overlap-mimi: how the Mimi stream learns that a chunk's codes are ready: a device-side wait_event (parks a blocked barrier packet in the second queue), or the host waits for the event and only then launches (one chunk behind the frame graphs)")
    ap.add_argument("--mimi-cus", type=int, default=0, help="with --overlap-mimi: restrict the Mimi stream to this many CUs (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--cu-pattern", default="low", choices=["low", "xcd"], help="which mask bits: the N lowest, or N/32 whole XCDs (bit i -> XCD i mod 8)")
    ap.add_argument("--lm-complement", action="store_true", help="restrict the frame-graph stream to the CUs the Mimi stream does not use")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"], help="weight format of the DualAR Linears (fp8 = e4m3 + row scales, BASELINE config 5; the model is then the dequantised one)")
    ap.add_argument("--streams", type=int, default=1, help="independent decode streams per GPU (slots are split evenly)")
    args = ap.parse_args()

    from smoltts_amd import parallel
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import NumericsMode, TokenConfig
    from smoltts_amd.engine import (EPI_SWIGLU, LMEngine, LMSession, MimiEngine, MimiSession, check,
                                    load_library)
    from smoltts_amd.packing import pack_lm, pack_mimi
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    torch.set_num_threads(host_threads())
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("SMOLTTS_BENCH_ONE_DEVICE") == "1":  # rehearsal of the N > 1 path on a 1-GPU box (with SMOLTTS_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = load_library()

    B, CH, K, W = args.batch, args.chunk, args.steps, args.warmup
    cfg = named_config(args.model)
    tok = load_tokenizer()
    tc = TokenConfig.from_tokenizer(tok, cfg)
    numerics = NumericsMode.torch_reference()
    # a slot's context may not outgrow the model's RoPE table: longest prompt (T <= 176 for these prompts) + all frames.
    # Many steps => fewer frames per step (logged; the JSON line names the chunk actually used).
    ch_max = (cfg.max_seq_len - 176 - 16) // max(W + K, 1)
    if ch_max < 1:
        raise SystemExit(f"--steps {K} --warmup {W}: even one frame per step exceeds max_seq_len={cfg.max_seq_len}")
    if CH > ch_max:
        log(f"chunk {CH} -> {ch_max} frames per step so that {W + K} steps fit max_seq_len={cfg.max_seq_len}")
        CH = ch_max
    total_frames = 1 + (W + K) * CH + 4  # frame 0 from prefill, +4 for the in-situ kernel timing frames

    # ---- weights: rank 0 builds + packs, everyone receives them over RCCL
    state = mstate = None
    arena = offsets = m_arena = m_offsets = None
    if rank == 0:
        log(f"building synthetic {args.model} + Mimi weights ({torch.get_num_threads()} host threads)")
        state = synthetic_lm_state(cfg, seed=0)
        arena, offsets = pack_lm(cfg, state, numerics, args.weights)
        mstate = synthetic_mimi_state(seed=0)
        m_arena, m_offsets = pack_mimi(mstate, 8, max_positions=2 * total_frames + 16)
        log(f"packed: LM arena {arena.numel() / 1e6:.1f} MB, Mimi arena {m_arena.numel() / 1e6:.1f} MB")
    arena, offsets = parallel.broadcast_weights(arena, offsets, dev)
    m_arena, m_offsets = parallel.broadcast_weights(m_arena, m_offsets, dev)
    eng = LMEngine(cfg, None, tc, numerics, arena=arena, offsets=offsets)
    meng = MimiEngine(None, 8, window=0, arena=m_arena, offsets=m_offsets)

    # ---- inputs: utterance u -> rank u mod world
    pe = PromptEncoder(tok, tc.semantic_start_id, cfg.num_codebooks, cfg.duplicate_code_0)
    all_prompts = make_prompts(pe, B * world)
    mine = [all_prompts[u] for u in parallel.shard_utterances(B * world, rank, world)]
    max_T = max(p.shape[1] for p in mine)
    S = args.streams
    if B % S:
        raise SystemExit("--batch must be divisible by --streams")
    Bs = B // S
    groups = [mine[i * Bs:(i + 1) * Bs] for i in range(S)]
    streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(S)]  # latency-critical frame graphs
    sessions = [LMSession(eng, max_batch=Bs, max_seq=max_T + total_frames + 8, max_rows=sum(p.shape[1] for p in g),
                          max_frames=total_frames) for g in groups]
    msessions = [MimiSession(meng, max_batch=Bs, max_chunk_frames=CH) for _ in range(S)]
    pcms = [torch.zeros(Bs, total_frames * 1920, dtype=torch.float32, device=dev) for _ in range(S)]
    sess = sessions[0]

    # The Mimi decode of chunk i only needs the codes of chunk i, so it runs on its own stream behind an
    # event and overlaps the (latency-bound, few-CU) frame graphs of chunk i+1.
    mimi_streams = [torch.cuda.Stream(device=dev, priority=0) for _ in range(S)]
    if args.overlap_mimi and args.mimi_cus > 0:
        import ctypes

        hip = ctypes.CDLL("libamdhip64.so")
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (n_cu + 31) // 32
        if args.cu_pattern == "low":
            bits = [i < args.mimi_cus for i in range(n_cu)]
        else:
            bits = [(i % 8) < args.mimi_cus // (n_cu // 8) for i in range(n_cu)]

        def masked_stream(sel, prio):
            m = (ctypes.c_uint32 * words)()
            for i, b in enumerate(sel):
                if b:
                    m[i // 32] |= 1 << (i % 32)
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), words, m)
            if rc != 0:
                raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
            return torch.cuda.ExternalStream(h.value, device=dev)

        mimi_streams = [masked_stream(bits, 0) for _ in range(S)]
        if args.lm_complement:
            streams = [masked_stream([not b for b in bits], -1) for _ in range(S)]
        log(f"CU masks: Mimi stream on {sum(bits)} of {n_cu} CUs ({args.cu_pattern}); frame graphs on {'the complement' if args.lm_complement else 'all'}")

    host_wait = args.overlap_mimi and args.overlap_wait == "host" and not args.no_mimi
    behind = [None] * S  # host-wait mode: (event, chunk index) of the chunk whose Mimi decode has not been launched yet

    def flush(j):
        if behind[j] is not None:
            ev, i = behind[j]
            ev.synchronize()  # the codes of chunk i are there; the frame graphs of chunk i + 1 are already queued
            with torch.cuda.stream(mimi_streams[j]):
                msessions[j].decode_chunk(sessions[j].codes, i * CH, CH, pcms[j], code_offset=1)
            behind[j] = None

    def step(i):
        for j in range(S):
            with torch.cuda.stream(streams[j]):
                sessions[j].decode(CH)
                ev = torch.cuda.Event()
                ev.record(streams[j])
            if host_wait:
                flush(j)
                behind[j] = (ev, i)
            elif not args.no_mimi:
                with torch.cuda.stream(mimi_streams[j] if args.overlap_mimi else streams[j]):
                    if args.overlap_mimi:
                        mimi_streams[j].wait_event(ev)
                    msessions[j].decode_chunk(sessions[j].codes, i * CH, CH, pcms[j], code_offset=1)

    log(f"sessions ready (B={B}, max_seq={sess.max_seq}); prefill of {sum(p.shape[1] for p in mine)} prompt rows")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(S):
        with torch.cuda.stream(streams[j]):
            sessions[j].prefill(groups[j], stop_on_eos=False)
    torch.cuda.synchronize()
    prefill_ms = (time.perf_counter() - t0) * 1e3
    for j in range(S):
        with torch.cuda.stream(mimi_streams[j] if args.overlap_mimi else streams[j]):
            msessions[j].reset()
    for i in range(W):
        step(i)
    for j in range(S):
        flush(j)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        step(i)
    for j in range(S):
        flush(j)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    elapsed = parallel.all_reduce_max(time.perf_counter() - t0, dev)
    frames_done = parallel.all_reduce_sum(float(B * CH * K), dev)
    value = frames_done / elapsed
    log(f"timed {K} steps: {elapsed * 1e3:.1f} ms -> {value:.0f} frames/s")

    torch.cuda.synchronize()
    fetched = [x.fetch() for x in sessions]
    codes = np.concatenate([f[0] for f in fetched])
    n_frames = np.concatenate([f[1] for f in fetched])
    margin = np.concatenate([f[3] for f in fetched])
    pcm = torch.cat(pcms)
    assert int(n_frames.min()) == 1 + (W + K) * CH, (n_frames, 1 + (W + K) * CH)
    assert bool(torch.isfinite(pcm[:, : (W + K) * CH * 1920]).all())

    # ---- dominant kernel in situ: replay the frame graph with every w1|w3 GEMM launch issued twice
    #      (idempotent), HIP events around the replays on the launch stream; the extra time per extra
    #      launch is the kernel's duration inside the real frame (cache state, neighbours and all)
    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        n_per_frame = cfg.n_layer + cfg.n_fast_layer * cfg.max_fast_seqlen
        nfr = 2  # frames per measurement (the session was sized with 4 spare frames... keep within them)

        def timed_frames(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(streams[0]):
                sess.decode(0)  # (re)capture happens on the first real launch below
                a.record(streams[0])
                sess.decode(n)
                b.record(streams[0])
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3  # us

        torch.cuda.synchronize()
        # measure on a scratch session so that the benchmarked sessions keep their frame budget
        scratch = LMSession(eng, max_batch=Bs, max_seq=max_T + 64, max_rows=sum(p.shape[1] for p in groups[0]), max_frames=64)
        with torch.cuda.stream(streams[0]):
            scratch.prefill(groups[0], stop_on_eos=False)
            scratch.decode(8)
        torch.cuda.synchronize()
        sess_saved, sess = sess, scratch
        base = min(timed_frames(8) for _ in range(3))
        check(lib.smoltts_debug_duplicate(EPI_SWIGLU, 2 * cfg.intermediate_size), "debug_duplicate")
        check(lib.smoltts_session_drop_graph(scratch.handle), "drop_graph")
        with torch.cuda.stream(streams[0]):
            scratch.decode(1)
        dup = min(timed_frames(8) for _ in range(3))
        check(lib.smoltts_debug_duplicate(-1, 0), "debug_duplicate")
        check(lib.smoltts_session_drop_graph(scratch.handle), "drop_graph")
        sess = sess_saved
        scratch.close()
        avg_us = (dup - base) / (8 * n_per_frame)
        log(f"in-situ w1|w3 GEMM: frame graph {base / 8:.1f} us -> {dup / 8:.1f} us with {n_per_frame} duplicated launches: {avg_us:.2f} us/launch")
        # algorithmic bytes of one launch: bf16 w1|w3 tiles + X3 operand in (6 B/elem) + X3 h out + partial sums of squares
        wbytes = 1 if args.weights == "fp8" else 2
        bytes_alg = 2 * cfg.intermediate_size * (cfg.dim * wbytes + (4 if args.weights == "fp8" else 0)) + Bs * cfg.dim * 6 + Bs * cfg.intermediate_size * 6 + Bs * (cfg.dim // 16) * 4
        ach = bytes_alg / (avg_us * 1e-6) / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
        pmc = ROOT / "profiles" / "r01_pmc_w13.json"
        if pmc.exists() and args.model == "smoltts_byte_150m" and Bs == 32 and args.weights == "bf16":
            traffic = json.loads(pmc.read_text())["hbm_bytes_per_launch"]
        roofline = {"bound": "hbm", "kernel": "gemm3_kernel<1, 3, 3, 2, false> = MT 1, T 3, U 3, SwiGLU epilogue, bf16 weights (RMSNorm-scaled w1|w3 GEMM)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "avg_us": round(avg_us, 3), "launches_timed": 8 * n_per_frame * 3,
                    "bytes_per_launch": bytes_alg, "method": "graph replay with duplicated launches, HIP events"}

    # ---- CPU baseline: the oracle (fp32 torch eager) on the same weights/prompts, bounded sample
    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle.lm_oracle import LMOracle, OracleLMConfig
        from oracle.mimi_oracle import MimiDecodeOracle

        nF = args.cpu_frames
        log(f"CPU oracle sample: prefill {B} prompts + {nF + 1} frames on {torch.get_num_threads()} threads")
        ocfg, ostate = (cfg, state)
        if args.weights == "fp8":  # the engine computes the dequantised model: that is what the oracle gets
            from smoltts_amd.packing import fp8_reference_state

            ocfg, ostate = fp8_reference_state(cfg, state)
        orc = LMOracle(OracleLMConfig.from_dict(ocfg.__dict__), ostate, embed_mask=numerics.embed_mask, rope_bf16=numerics.rope_bf16)
        with torch.no_grad():
            orc._alloc(B, max_T + nF + 2)
            hidden = torch.stack([orc.prefill_one(b, torch.from_numpy(mine[b]).long()) for b in range(B)])
            cols_all = []
            t_lm = 0.0
            for f in range(nF + 1):  # frame 0 is the warm-up, untimed
                t1 = time.perf_counter()
                ids = orc.slow_head(hidden).argmax(-1)
                cds, _ = orc.fast_decode(hidden)
                cols = torch.cat([ids[:, None], cds], dim=1)
                hidden = orc.decode_cols(cols)
                if f > 0:
                    t_lm += time.perf_counter() - t1
                cols_all.append(cols)
                log(f"  oracle frame {f} done")
            grid = torch.stack(cols_all, dim=1)  # B, nF+1, 9
            morc = MimiDecodeOracle(mstate)
            t1 = time.perf_counter()
            ref_pcm = morc.decode(grid[:, 1:, 1:].permute(0, 2, 1).contiguous())
            t_mimi = time.perf_counter() - t1
            log(f"  oracle Mimi decode done ({t_mimi:.1f}s)")
        cpu = {"value": round(B * nF / (t_lm + t_mimi), 2), "unit": "frames/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"{nF} decode frames x {B} utterances (150m oracle, fp32 torch eager, KV-cached) "
               f"+ Mimi decode of those frames; LM {t_lm:.2f}s, Mimi {t_mimi:.2f}s; prefill and 1 warm-up frame untimed"}
        same = np.array_equal(codes[:, : nF + 1], grid.numpy())
        pcm_ref_full = MimiDecodeOracle(mstate).decode(grid[:, :, 1:].permute(0, 2, 1).contiguous())[:, 0].numpy()
        rms = float(np.sqrt(np.mean((pcm[:, : (nF + 1) * 1920].cpu().numpy() - pcm_ref_full) ** 2)))
        parity = {"frames_checked": nF + 1, "ids_bit_identical": bool(same), "pcm_rms_err": rms}

    # ---- first-audio-chunk latency (second half of BASELINE.json's metric): submit -> first 1920 PCM
    #      samples on the host = prompt prefill + frame 0 + one Mimi step.  (a) this workload: all B
    #      utterances submitted together; (b) BASELINE configs[1]: smoltts_byte_70m, one utterance.
    first_chunk = None
    if rank == 0 and world == 1 and not args.no_latency:
        def first_chunk_ms(engine, mengine, prompts, reps):
            ts = []
            ls = LMSession(engine, max_batch=len(prompts), max_seq=max(p.shape[1] for p in prompts) + 8,
                           max_rows=sum(p.shape[1] for p in prompts), max_frames=4)
            ms = MimiSession(mengine, max_batch=len(prompts), max_chunk_frames=1)
            buf = torch.zeros(len(prompts), 1920, dtype=torch.float32, device=dev)
            for _ in range(reps + 1):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ls.prefill(prompts, stop_on_eos=False)
                ms.reset()
                ms.decode_chunk(ls.codes, 0, 1, buf, code_offset=1)
                _ = buf.cpu()
                ts.append((time.perf_counter() - t1) * 1e3)
            ms.close(); ls.close()
            return ts[1:]  # first repetition warms the allocator

        t150 = first_chunk_ms(eng, meng, mine, 5)

        def steady_first_chunk_ms(engine, mengine, prompts, reps):
            """Continuous batching: all B slots are speaking; one slot is restarted with a new prompt (prefill of that
            prompt + frame 0) and its first frame goes through a one-slot Mimi session to the host."""
            ls = LMSession(engine, max_batch=len(prompts), max_seq=max(p.shape[1] for p in prompts) + 64,
                           max_rows=sum(p.shape[1] for p in prompts), max_frames=64)
            ls.prefill(prompts, stop_on_eos=False)
            ls.decode(4)
            ms1 = MimiSession(mengine, max_batch=1, max_chunk_frames=1)
            buf = torch.zeros(1, 1920, dtype=torch.float32, device=dev)
            ts = []
            for i in range(reps + 2):
                k = i % len(prompts)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ls.prefill([prompts[(k + 7) % len(prompts)]], slots=[k], stop_on_eos=False)
                ms1.reset()
                ms1.decode_chunk(ls.codes[k:k + 1], 0, 1, buf, code_offset=1)
                _ = buf.cpu()
                ts.append((time.perf_counter() - t1) * 1e3)
                ls.decode(1)  # the other slots keep speaking between arrivals
            ms1.close(); ls.close()
            return ts[2:]

        t150s = steady_first_chunk_ms(eng, meng, mine, 30)
        cfg70 = named_config("smoltts_byte_70m")
        a70, o70 = pack_lm(cfg70, synthetic_lm_state(cfg70, seed=0), numerics)
        eng70 = LMEngine(cfg70, None, TokenConfig.from_tokenizer(tok, cfg70), numerics, arena=a70, offsets=o70)
        t70 = []
        for u in range(200):  # p50 / p95 over 200 prompts
            t70 += first_chunk_ms(eng70, meng, [all_prompts[u % len(all_prompts)]], 1)
        # BASELINE configs[1]: one 70m stream, every frame decoded to PCM and copied to the host as it appears
        ls1 = LMSession(eng70, max_batch=1, max_seq=400, max_rows=256, max_frames=160)
        ms1 = MimiSession(meng, max_batch=1, max_chunk_frames=1)
        buf1 = torch.zeros(1, 1920, dtype=torch.float32, device=dev)
        ls1.prefill([all_prompts[0]], stop_on_eos=False)
        ms1.reset()
        for f in range(150):
            if f == 22:  # the first frames warm the graph and the allocator
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            if f:
                ls1.decode(1)
            ms1.decode_chunk(ls1.codes, f, 1, buf1, code_offset=1)
            _ = buf1.cpu()
        b1_stream_fps = 128 / (time.perf_counter() - t1)
        ms1.close(); ls1.close()
        first_chunk = {"b1_70m_stream_frames_per_s": round(b1_stream_fps, 1),
                       "b32_150m_ms_p50": round(float(np.median(t150)), 2),
                       "b32_150m_steady_ms_p50": round(float(np.median(t150s)), 2), "b32_150m_steady_ms_p95": round(float(np.percentile(t150s, 95)), 2),
                       "b1_70m_ms_p50": round(float(np.median(t70)), 2), "b1_70m_ms_p95": round(float(np.percentile(t70, 95)), 2),
                       "b1_70m_prompts": len(t70), "includes": "prefill + frame 0 + Mimi step + D2H copy of 1920 samples; b32_150m: all 32 prompts submitted together; "
                                   "b32_150m_steady: one new prompt into a session whose 32 slots are all speaking"}
        log(f"first audio chunk: B=32 150m p50 {first_chunk['b32_150m_ms_p50']} ms (all at once) / {first_chunk['b32_150m_steady_ms_p50']} ms "
            f"(one arrival among 32 speaking slots); B=1 70m p50 {first_chunk['b1_70m_ms_p50']} ms")

    if rank == 0:
        out = {
            "metric": "Mimi frames/sec (=12.5 x RTF), greedy DualAR decode + Mimi decode to PCM" + (" [DIAGNOSTIC: Mimi skipped]" if args.no_mimi else ""),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": f"{'fp8-e4m3 (row-scaled)' if args.weights == 'fp8' else 'bf16'} weights, fp32 activations/accumulate (LM); fp32 (Mimi)",
            "data": "synthetic (seeded random weights at the real shapes, synthetic ChatML prompts)",
            "config": {"workload": f"{args.model} B={B}/GPU concurrent utterances, chunk {CH} frames/step, "
                                   f"prompts T={min(p.shape[1] for p in mine)}..{max_T}, context {max_T + W * CH}..{max_T + (W + K) * CH}",
                       "global_batch": B * world, "frames_per_step": B * CH * world, "parallelism": f"dp{world} (utterance-sharded replicas)", "streams_per_gpu": S},
            "rtf": round(value / 12.5, 1), "frames_per_s_per_gpu": round(value / world, 1),
            "us_per_frame_step": round(elapsed / (K * CH) * 1e6, 1), "prefill_ms": round(prefill_ms, 2),
            "min_top2_margin": float(margin.min()),
            "first_audio_chunk": first_chunk, "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        }
        print(json.dumps(out), flush=True)
    for x in msessions + sessions:
        x.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
