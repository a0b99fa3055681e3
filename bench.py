#!/usr/bin/env python3
"""Throughput bench of the hot path: greedy DualAR decode + Mimi decode to PCM.

Metric (BASELINE.json): Mimi frames/s (= 12.5 x real-time factor), whole job over all GPUs.
Workload at N=1 (BASELINE.json configs[2]): smoltts_byte_150m, bf16 weights, B=32 concurrent
utterances, synthetic ChatML prompts, seeded random weights, greedy, EOS disabled.

One *step* = one chunk of CH frames for all B slots of the rank: CH replays of the captured
frame graph (slow step + 8 depth steps + on-device argmax each) followed by one Mimi chunk decode
of those CH x B frames to PCM, everything resident in HBM.  Prompt prefill happens before the
timed region (the reference's own "x realtime" excludes it, lm/generate.py:199-214) and is
reported separately.

N > 1 (BASELINE.json configs[3]): one process per GPU, utterances sharded by rank (weak scaling, no
per-step collective), weights broadcast from rank 0 over RCCL.  `python bench.py --gpus N` starts its
own N ranks (the parent never touches a GPU: it only spawns, relays rank 0's line and returns the
children's exit code); under `python -m torch.distributed.run ... bench.py --gpus N` it is a rank.

Other documented invocations:
  config 5 (fp8 weights + chunked prompt prefill, B=64):  python bench.py --weights fp8 --batch 64 --prefill-chunk 128
  launcher rehearsal without a GPU (gloo, no compute):    python bench.py --gpus 2 --rehearse-launcher

Output: ONE JSON line on rank 0 (driver contract) with `roofline` (dominant kernel: the fused
RMSNorm + w1|w3 GEMM + SwiGLU, timed in situ with HIP events; plus the whole frame-step against
SURVEY.md §8d's algorithmic bytes) and `cpu_baseline` (the CPU oracle, fp32 torch eager, timed on
this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PROFILE_TAG = "r04"    # profiles/<tag>_* hold the rocprofv3 summaries of this round

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
    import numpy as np

    from smoltts_amd.prompt import VOICES

    rng = np.random.default_rng(seed)
    out = []
    for u in range(n_total):
        n = int(rng.integers(40, 161))
        text = "".join(chr(int(c)) for c in rng.integers(32, 127, size=n))
        out.append(pe.build_prompt(text, VOICES[u % len(VOICES)]))
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="smoltts_byte_150m")
    ap.add_argument("--batch", type=int, default=32, help="utterance slots per GPU")
    ap.add_argument("--chunk", type=int, default=32, help="frames per step")
    ap.add_argument("--cpu-frames", type=int, default=32, help="frames of the CPU-oracle sample (0 = skip; SURVEY.md §8d: 32)")
    ap.add_argument("--prefill-chunk", type=int, default=0, help="prompt columns per utterance per prefill call (0 = whole prompts; BASELINE config 5: 128)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the first-audio-chunk latency measurement")
    ap.add_argument("--no-mimi", action="store_true", help="diagnostic only: skip the Mimi decode (the result line is then not the metric)")
    ap.add_argument("--overlap-mimi", dest="overlap_mimi", action="store_true", default=True,
                    help="(default) the Mimi chunk decode of chunk i runs on a second stream beside the frame graphs of chunk i + 1, launched by the host once it "
                         "has seen chunk i's codes (what the serving loop does with its codec passes); measured +3.5 %% at the round-4 kernels "
                         "(profiles/r04_ab_codec_overlap.txt)")
    ap.add_argument("--no-overlap-mimi", dest="overlap_mimi", action="store_false", help="the Mimi chunk decode on the frame graphs' own stream, after them (rounds 1-3; what the profiler scripts use)")
    ap.add_argument("--overlap-wait", default="host", choices=["device", "host"], help="with --overlap-mimi: how the Mimi stream learns that a chunk's codes are ready: the host waits for the event and only then launches (one chunk behind the frame graphs), or a device-side wait_event (parks a blocked barrier packet in the second queue: every dependent launch of the frame graphs slows, -13 %%)")
    ap.add_argument("--mimi-cus", type=int, default=128, help="with --overlap-mimi: restrict the Mimi stream to this many CUs (hipExtStreamCreateWithCUMask; 0 = no mask): at 128 of 256 the frame graphs always find half the chip free (measured 32.1k against 31.4k unmasked and 31.3k at 64)")
    ap.add_argument("--cu-pattern", default="low", choices=["low", "xcd"], help="which mask bits: the N lowest, or N/32 whole XCDs (bit i -> XCD i mod 8)")
    ap.add_argument("--lm-complement", action="store_true", help="restrict the frame-graph stream to the CUs the Mimi stream does not use")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"], help="weight format of the DualAR Linears (fp8 = e4m3 storage + row scales, dequantised to bf16 in registers: the MFMA operands stay bf16; BASELINE config 5; the model is then the dequantised one)")
    ap.add_argument("--fp8-prefill", action="store_true", help="with --weights fp8: prompt prefills of >= 256 rows on the fp8 x fp8 MFMA (SMOLTTS_OPT_FP8_PREFILL: BASELINE configs[4]'s 'fp8 MFMA prefill'); the prompt KV rows are then approximate, ids are not comparable with the oracle: the oracle legs are skipped and the line says so")
    ap.add_argument("--codec-products", type=int, default=6, choices=[3, 6], help="bf16x3 products per operand pair in the codec's matrix-core kernels (SMOLTTS_MIMI_OPT_PRODUCTS): 6 = fp32-grade (default, PCM RMS ~1e-7 vs the fp32 oracle), 3 = the 2^-16-grade form (chunks 23 %% faster, PCM RMS ~7e-7; the bar is 1e-4) -- the line's dtype says which")
    ap.add_argument("--streams", type=int, default=1, help="independent decode streams per GPU (slots are split evenly)")
    ap.add_argument("--kv", default="fp32", choices=["fp32", "bf16"], help="KV-cache storage of the slow transformer (bf16: K/V rounded once when written; the oracle rounds identically)")
    ap.add_argument("--rehearse-launcher", action="store_true", help="no GPU, no compute: start the ranks, run the distributed plumbing of the bench (rendezvous, ranks_seen all-reduce, weight broadcast, sharding, barriers, timing reductions) over gloo and print the line")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------- launcher
def _launch_once(n: int, argv, pin: bool, extra_env=None):
    """One attempt of the parent of an N-rank run (-> (exit code, seconds it took, did rank 0 print its line, the rank that failed
    first or -1); a rank killed by a signal has a NEGATIVE code, as subprocess reports it): spawn one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run sets them), relay rank 0's stdout, return the first non-zero exit code (0 if none).  This
    process makes no GPU call (it does not even import torch).  Every child runs in a session of its own and sees exactly
    one GPU (HIP_VISIBLE_DEVICES = its entry of the parent's device list, set before the child's first GPU call;
    SMOLTTS_BENCH_PIN_DEVICES=0 leaves the binding to LOCAL_RANK -> torch.cuda.set_device as under torch.distributed.run).
    A failing rank takes the others down with it, and so does a signal to the parent (SIGTERM / SIGINT from a driver's
    timeout or Ctrl-C): children are terminated, then killed, by PID -- none is left holding a GPU."""
    import signal
    import socket
    import subprocess
    import threading

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    t_start = time.time()
    printed = []
    visible = [d for d in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if d.strip()] or [str(i) for i in range(n)]
    if pin and len(visible) < n:
        print(f"[bench launcher] HIP_VISIBLE_DEVICES lists {len(visible)} devices for {n} ranks", file=sys.stderr, flush=True)
        return 2, 0.0, False, -1
    procs = []
    failed_rank = -1

    def stop_children(grace=15.0):
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_end = time.time() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    got_signal = []

    def on_signal(signum, _frame):
        got_signal.append(signum)

    old_handlers = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    t = None
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
            env.update(extra_env or {})
            if pin:
                env["HIP_VISIBLE_DEVICES"] = visible[r]
                env["SMOLTTS_BENCH_PINNED"] = "1"  # the rank's one visible GPU is device 0
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env, start_new_session=True,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=(r == 0) or None))

        def relay(p):  # the result line goes to stdout; anything else a library prints there (gloo / RCCL banners) to stderr
            for line in p.stdout:
                is_line = line.lstrip().startswith("{")
                if is_line:
                    printed.append(1)
                dst = sys.stdout if is_line else sys.stderr
                dst.write(line)
                dst.flush()

        t = threading.Thread(target=relay, args=(procs[0],), daemon=True)
        t.start()
        live = set(range(n))
        while live and rc == 0 and not got_signal:
            time.sleep(0.05)
            for r in list(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc, failed_rank = code, r
                        print(f"[bench launcher] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
        if got_signal:
            rc = 128 + got_signal[0]
            print(f"[bench launcher] signal {got_signal[0]}: stopping the ranks", file=sys.stderr, flush=True)
    finally:
        stop_children()  # (a no-op when every rank has exited by itself)
        for sig, h in old_handlers.items():
            signal.signal(sig, h)
        if t is not None:
            t.join(timeout=5)
    return rc, time.time() - t_start, bool(printed), failed_rank


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N`: start the N ranks (see _launch_once).  Ranks are pinned to one GPU each through
    HIP_VISIBLE_DEVICES; should that attempt die before rank 0 has printed anything (a collective backend that cannot cope with
    per-process device masks shows at the rendezvous / weight broadcast, in the first seconds), ONE second attempt runs with
    the binding left to LOCAL_RANK -> torch.cuda.set_device, as under torch.distributed.run -- said loudly on stderr."""
    pin = os.environ.get("SMOLTTS_BENCH_PIN_DEVICES", "1") != "0" and os.environ.get("SMOLTTS_BENCH_ONE_DEVICE") != "1"
    rc, took, printed, failed = _launch_once(n, argv, pin)
    # Only an ordinary error exit (0 < rc < 128: an exception at the rendezvous / collective set-up) is retried.  A rank killed by
    # a signal (negative code: SIGSEGV, SIGABRT -- a GPU memory fault, an abort inside a library) is NOT re-run on the box: the run
    # fails with 128 + signal so that its cause is looked for in the records.
    if 0 < rc < 128 and pin and not printed and os.environ.get("SMOLTTS_BENCH_NO_FALLBACK") != "1":
        print(f"[bench launcher] the pinned attempt failed with {rc} (rank {failed}) after {took:.0f} s before any result: "
              "retrying ONCE without per-rank HIP_VISIBLE_DEVICES masks (binding by LOCAL_RANK)", file=sys.stderr, flush=True)
        note = json.dumps({"rc": rc, "rank": failed, "seconds": round(took, 1)})  # lands on the result line: launcher_fallback / pinned_attempt_failed
        rc, _, _, _ = _launch_once(n, argv, False, {"SMOLTTS_BENCH_PINNED_ATTEMPT_FAILED": note})
    if rc < 0:
        print(f"[bench launcher] rank {failed} was killed by signal {-rc}: no retry", file=sys.stderr, flush=True)
        rc = 128 - rc
    return rc


def launcher_record():
    """What the self-launcher did before this attempt, for the result line: (launcher_fallback, pinned_attempt_failed)."""
    note = os.environ.get("SMOLTTS_BENCH_PINNED_ATTEMPT_FAILED")
    return (True, json.loads(note)) if note else (False, None)


def rehearse(args) -> None:
    """The distributed plumbing of the bench without a GPU and without compute (gloo): what `--gpus N` adds to the
    single-GPU run, exercised end to end through the same launcher and the same smoltts_amd.parallel calls."""
    import torch

    from smoltts_amd import parallel
    from smoltts_amd.config import NumericsMode
    from smoltts_amd.packing import pack_lm
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    rank, world, _ = parallel.init_distributed(backend="gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("SMOLTTS_BENCH_FAIL_RANK") == str(rank):  # test hook of the launcher: a rank that dies early
        raise SystemExit(3)
    if os.environ.get("SMOLTTS_BENCH_FAIL_IF_PINNED") == "1" and os.environ.get("SMOLTTS_BENCH_PINNED") == "1" and rank == 1:
        raise SystemExit(5)  # test hook: a backend that cannot work under per-rank device masks
    if os.environ.get("SMOLTTS_BENCH_FAIL_IF_PINNED") == "abort" and os.environ.get("SMOLTTS_BENCH_PINNED") == "1" and rank == 1:
        os.abort()  # test hook: a rank that dies of a signal at start-up (what a GPU fault looks like to the launcher)
    seen = parallel.ranks_seen("cpu")
    cfg = named_config("tiny")
    arena = offsets = None
    if rank == 0:
        arena, offsets = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), NumericsMode.torch_reference())
    arena, offsets = parallel.broadcast_weights(arena, offsets, torch.device("cpu"))
    B = args.batch
    mine = parallel.shard_utterances(B * world, rank, world)
    parallel.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)  # stands in for the timed steps
    parallel.barrier()
    elapsed = parallel.all_reduce_max(time.perf_counter() - t0, "cpu")
    units = parallel.all_reduce_sum(float(len(mine)), "cpu")
    same_arena = parallel.identical_on_all_ranks(parallel.arena_checksum(arena), "cpu")
    if not same_arena:
        raise SystemExit(f"rank {rank}: the weight arena differs between ranks after the broadcast")
    # stands in for the per-rank golden decode of the real run: every rank checks something of its own and the passes are summed
    passed = parallel.all_reduce_sum(1.0 if int(arena.numel()) > 0 and len(mine) > 0 else 0.0, "cpu")
    if os.environ.get("SMOLTTS_BENCH_SLEEP_RANKS"):  # test hook of the launcher: ranks that outlive a signalled parent would show
        time.sleep(float(os.environ["SMOLTTS_BENCH_SLEEP_RANKS"]))
    if rank == 0:
        print(json.dumps({"metric": "REHEARSAL of the N-rank launcher and collectives (gloo, CPU, no compute): not a measurement",
                          "value": None, "unit": "frames/s", "n_gpus": world, "ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
                          "utterances_sharded": int(units), "arena_bytes": int(arena.numel()), "arena_identical_on_all_ranks": same_arena,
                          "parity": {"ranks_checked": world, "ranks_passed": int(passed), "arena_checksums_identical": same_arena},
                          "elapsed_s": round(elapsed, 4), "backend": "gloo", "launcher_fallback": launcher_record()[0],
                          "pinned_attempt_failed": launcher_record()[1]}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


# ------------------------------------------------------------------------------------------- one rank
def step_bytes(cfg, B, L, ch, weights="bf16"):
    """algorithmic HBM bytes of one frame-step (one frame for all B utterances of a GPU):
    W_slow + n_fast * (W_fast + W_head_slice) + B * L * KV_tok (bf16-algorithmic: 2 * kvh * 64 * 2 B * layers) + the Mimi
    decode of those frames (fp32 weights streamed once per chunk of `ch` frames + ~2.5 MB of activations per frame)."""
    wb = 1 if weights == "fp8" else 2

    def block(d, h, kv, inter):
        return ((h + 2 * kv) * 64 * d + d * d + 2 * inter * d + d * inter) * wb + 2 * d * 4

    w_slow = cfg.n_layer * block(cfg.dim, cfg.n_head, cfg.n_local_heads, cfg.intermediate_size) + cfg.vocab_size * cfg.dim * wb
    w_fast = cfg.n_fast_layer * block(cfg.fast_dim, cfg.fast_n_head, cfg.fast_n_local_heads, cfg.fast_intermediate_size)
    w_head = cfg.fast_dim * cfg.codebook_size * wb
    kv_tok = 2 * cfg.n_local_heads * 64 * 2 * cfg.n_layer
    mimi_w = 100.7e6 + 58.9e6  # decoder transformer + SEANet, fp32
    parts = {"lm_weights": w_slow + cfg.max_fast_seqlen * (w_fast + w_head), "kv": B * L * kv_tok,
             "mimi_weights_per_chunk_share": mimi_w / ch, "mimi_activations": B * 2.5e6}
    return sum(parts.values()), parts


def run_rank(args) -> None:
    import numpy as np
    import torch

    from smoltts_amd import parallel
    from smoltts_amd.codec.synthetic import synthetic_mimi_state
    from smoltts_amd.config import NumericsMode, TokenConfig
    from smoltts_amd.engine import EPI_SWIGLU, LMEngine, LMSession, MimiEngine, MimiSession, load_library
    from smoltts_amd.packing import pack_lm, pack_mimi
    from smoltts_amd.prompt import PromptEncoder
    from smoltts_amd.synthetic import named_config, synthetic_lm_state
    from smoltts_amd.tokenizer import load_tokenizer

    torch.set_num_threads(host_threads())
    # SMOLTTS_BENCH_ONE_DEVICE: rehearsal of the N > 1 path on a 1-GPU box (with SMOLTTS_DIST_BACKEND=gloo);
    # SMOLTTS_BENCH_PINNED: started by self_launch under a one-device HIP_VISIBLE_DEVICES mask -- either way the rank's GPU is device 0
    one_dev = os.environ.get("SMOLTTS_BENCH_ONE_DEVICE") == "1" or os.environ.get("SMOLTTS_BENCH_PINNED") == "1"
    rank, world, local = parallel.init_distributed(device_index=0 if one_dev else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    load_library()
    seen = parallel.ranks_seen(dev)  # over RCCL: every rank's GPU contributes a 1
    if seen != world:
        raise SystemExit(f"the collective backend connects {seen} ranks, WORLD_SIZE is {world}")

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
    total_frames = 1 + (W + K) * CH + 4  # frame 0 from prefill, + spare

    # ---- weights: rank 0 builds + packs, everyone receives them over RCCL
    state = mstate = None
    arena = offsets = m_arena = m_offsets = None
    if rank == 0:
        log(f"building synthetic {args.model} + Mimi weights ({torch.get_num_threads()} host threads)")
        state = synthetic_lm_state(cfg, seed=0)
        arena, offsets = pack_lm(cfg, state, numerics, args.weights)
        mstate = synthetic_mimi_state(seed=0)
        m_arena, m_offsets = pack_mimi(mstate, 8, max_positions=2 * max(total_frames, 160) + 16)  # (the latency probe streams 150 frames)
        log(f"packed: LM arena {arena.numel() / 1e6:.1f} MB, Mimi arena {m_arena.numel() / 1e6:.1f} MB")
    arena, offsets = parallel.broadcast_weights(arena, offsets, dev)
    m_arena, m_offsets = parallel.broadcast_weights(m_arena, m_offsets, dev)
    # every rank must hold the same bytes: checksums of both arenas, all-gathered and compared on every rank
    arenas_same = parallel.identical_on_all_ranks(parallel.arena_checksum(arena) + parallel.arena_checksum(m_arena), dev)
    if not arenas_same:
        raise SystemExit(f"rank {rank}: the weight arenas differ between ranks after the broadcast")
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
    pch = args.prefill_chunk
    sessions = [LMSession(eng, max_batch=Bs, max_seq=max_T + total_frames + 8,
                          max_rows=sum(min(p.shape[1], pch) if pch else p.shape[1] for p in g), max_frames=total_frames, kv_dtype=args.kv)
                for g in groups]
    msessions = [MimiSession(meng, max_batch=Bs, max_chunk_frames=CH, products=args.codec_products) for _ in range(S)]
    pcms = [torch.zeros(Bs, total_frames * 1920, dtype=torch.float32, device=dev) for _ in range(S)]
    sess = sessions[0]

    def do_prefill(session, prompts, **kw):
        if pch:
            session.prefill_chunked(prompts, chunk=pch, **kw)
        else:
            session.prefill(prompts, **kw)

    # The Mimi decode of chunk i only needs the codes of chunk i, so it runs on its own stream beside the frame graphs of chunk
    # i + 1 (--overlap-mimi, the default since round 4; --no-overlap-mimi puts it back on the frame graphs' stream).
    mimi_streams = [torch.cuda.Stream(device=dev, priority=0) for _ in range(S)]
    mimi_cus_used = None
    if args.overlap_mimi and args.mimi_cus > 0 and not args.no_mimi:
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
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
            return torch.cuda.ExternalStream(h.value, device=dev)

        try:
            mimi_streams = [masked_stream(bits, 0) for _ in range(S)]
            mimi_cus_used = sum(bits)
        except (RuntimeError, OSError, AttributeError) as e:  # no masked stream on this runtime: the plain second stream (said on the result line: no "-CU mask")
            if args.lm_complement:
                raise SystemExit(str(e))
            log(f"no CU-masked stream ({e}): the Mimi stream runs unmasked")
        if args.lm_complement:
            streams = [masked_stream([not b for b in bits], -1) for _ in range(S)]
        if mimi_cus_used:
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

    log(f"sessions ready (B={B}, max_seq={sess.max_seq}, KV {args.kv}); prefill of {sum(p.shape[1] for p in mine)} prompt rows"
        + (f" in chunks of {pch} columns" if pch else ""))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(S):
        with torch.cuda.stream(streams[j]):
            do_prefill(sessions[j], groups[j], stop_on_eos=False)
    torch.cuda.synchronize()
    prefill_ms = (time.perf_counter() - t0) * 1e3
    for j in range(S):  # frames per graph launch: chosen here (and captured now), not by whichever decode call comes first
        with torch.cuda.stream(streams[j]):
            sessions[j].set_frames_per_graph(min(CH, 8))
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
    mine_s = time.perf_counter() - t0  # this rank's own time for its K steps (reporting only)
    parallel.barrier()
    torch.cuda.synchronize()
    elapsed = parallel.all_reduce_max(time.perf_counter() - t0, dev)
    frames_done = parallel.all_reduce_sum(float(B * CH * K), dev)
    rank_fps = parallel.all_gather_floats(B * CH * K / mine_s, dev)
    value = frames_done / elapsed
    log(f"timed {K} steps: {elapsed * 1e3:.1f} ms -> {value:.0f} frames/s")

    torch.cuda.synchronize()
    fetched = [x.fetch() for x in sessions]
    codes = np.concatenate([f[0] for f in fetched])
    n_frames = np.concatenate([f[1] for f in fetched])
    margin = np.concatenate([f[3] for f in fetched])
    margin_at = np.concatenate([x.margin_at.cpu().numpy() for x in sessions])
    pcm = torch.cat(pcms)
    n_total = 1 + (W + K) * CH
    assert int(n_frames.min()) == n_total, (n_frames, n_total)
    if not args.no_mimi:
        assert bool(torch.isfinite(pcm[:, : (W + K) * CH * 1920]).all())
    us_per_frame_step = elapsed / (K * CH) * 1e6
    L_mean = float(np.mean([p.shape[1] for p in mine])) + (W + K / 2) * CH  # mean context over the timed steps

    # ---- every rank proves its own GPU on the committed golden grid (made with the reference model in the build container,
    #      tests/golden/make_lm_goldens.py): prompt_0 of lm_150m.npz decoded for its 16 frames must equal grid_0 bit for bit.
    #      No CPU oracle involved, so it runs on all ranks of an N > 1 job; the passes are summed over the backend.
    golden_ok = None
    gpath = ROOT / "tests" / "golden" / "lm_150m.npz"
    if args.model == "smoltts_byte_150m" and args.weights == "bf16" and args.kv == "fp32" and gpath.exists():
        g = np.load(gpath)
        assert int(g["seed"]) == 0 and str(g["config_name"]) == args.model
        gf = int(g["frames"])
        gs = LMSession(eng, max_batch=1, max_seq=int(g["prompt_0"].shape[1]) + gf + 8, max_rows=int(g["prompt_0"].shape[1]), max_frames=gf)
        with torch.cuda.stream(streams[0]):
            gs.prefill([g["prompt_0"]], stop_on_eos=False)
            gs.decode(gf - 1)
        torch.cuda.synchronize()
        gcodes, gn, _, _ = gs.fetch()
        golden_ok = bool(int(gn[0]) == gf and np.array_equal(gcodes[0, :gf].T, g["grid_0"]))
        gs.close()
    ranks_checked = int(parallel.all_reduce_sum(0.0 if golden_ok is None else 1.0, dev))
    ranks_passed = int(parallel.all_reduce_sum(1.0 if golden_ok else 0.0, dev))
    rank_parity = {"ranks_checked": ranks_checked, "ranks_passed": ranks_passed, "arena_checksums_identical": arenas_same,
                   "golden": "tests/golden/lm_150m.npz: prompt_0 decoded for 16 frames on every rank == grid_0 (ids bit-identical)" if ranks_checked else None}
    log(f"golden grid on every rank: {ranks_passed} of {ranks_checked} ranks reproduce it; arenas identical on all ranks: {arenas_same}")

    # ---- dominant kernel in situ: replay the frame graph with every w1|w3 GEMM launch issued twice
    #      (idempotent), HIP events around the replays on the launch stream; the extra time per extra
    #      launch is the kernel's duration inside the real frame (cache state, neighbours and all)
    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        n_per_frame = cfg.n_layer + cfg.n_fast_layer * cfg.max_fast_seqlen

        def timed_frames(s_, n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(streams[0]):
                s_.decode(n)  # both graphs (one frame, n frames) are (re)captured and warm here, outside the events
                a.record(streams[0])
                s_.decode(n)
                b.record(streams[0])
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3  # us

        torch.cuda.synchronize()
        # measure on a scratch session at the run's mean context so that the benchmarked sessions keep their frame budget
        ctx = int((W + K / 2) * CH)
        scratch = LMSession(eng, max_batch=Bs, max_seq=max_T + ctx + 128, max_rows=sum(p.shape[1] for p in groups[0]), max_frames=ctx + 120,
                            kv_dtype=args.kv)
        with torch.cuda.stream(streams[0]):
            scratch.prefill(groups[0], stop_on_eos=False)
            scratch.decode(ctx)
        torch.cuda.synchronize()
        base = min(timed_frames(scratch, 8) for _ in range(3))  # (6 x 16 frames in all: inside the scratch session's budget)
        scratch.measure_duplicate(EPI_SWIGLU, 2 * cfg.intermediate_size)  # this session only; its graphs are dropped
        dup = min(timed_frames(scratch, 8) for _ in range(3))
        scratch.measure_duplicate(-1)
        scratch.close()
        avg_us = (dup - base) / (8 * n_per_frame)
        log(f"in-situ w1|w3 GEMM at context ~{max_T + ctx}: frame graph {base / 8:.1f} us -> {dup / 8:.1f} us with {n_per_frame} duplicated launches: {avg_us:.2f} us/launch")
        # algorithmic bytes of one launch: bf16 w1|w3 tiles + X3 operand in (6 B/elem) + X3 h out + partial sums of squares
        wbytes = 1 if args.weights == "fp8" else 2
        bytes_alg = 2 * cfg.intermediate_size * (cfg.dim * wbytes + (4 if args.weights == "fp8" else 0)) + Bs * cfg.dim * 6 + Bs * cfg.intermediate_size * 6 + Bs * (cfg.dim // 16) * 4
        ach = bytes_alg / (avg_us * 1e-6) / 1e9
        # HBM traffic and the profiler's own average for this kernel come from the committed rocprofv3 summaries of this
        # same command (they cannot be collected from inside the process): labelled with their source
        traffic = traffic_src = rocprof_us = rocprof_src = None
        pmc = ROOT / "profiles" / f"{PROFILE_TAG}_pmc_w13.json"
        std = args.model == "smoltts_byte_150m" and Bs == 32 and args.weights == "bf16" and args.kv == "fp32"
        if pmc.exists() and std:
            j = json.loads(pmc.read_text())
            traffic, traffic_src = j["hbm_bytes_per_launch"], f"profiles/{pmc.name}: {j.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}"
        kst = ROOT / "profiles" / f"{PROFILE_TAG}_bench_kernel_stats.csv"
        if kst.exists() and std:
            import csv

            calls = tot_ns = 0.0  # two instantiations since round 3: the slow layers' (weights streamed past the caches) and the depth layers'
            for row in csv.DictReader(kst.open()):
                if "gemm3_kernel<1, 3, 3, 2" in row.get("Name", ""):
                    calls += float(row["Calls"])
                    tot_ns += float(row["Calls"]) * float(row["AverageNs"])
            if calls:
                rocprof_us = round(tot_ns / calls / 1e3, 3)
                rocprof_src = f"profiles/{kst.name} (rocprofv3 --kernel-trace --stats of `python3 bench.py --cpu-frames 0 --no-latency --no-overlap-mimi`, profiler attached; both instantiations of the kernel, weighted by calls)"
        sb, parts = step_bytes(cfg, B, L_mean, CH, args.weights)
        step_ach = sb / (us_per_frame_step * 1e-6) / 1e9
        roofline = {"bound": "hbm", "kernel": "gemm3_kernel<1, 3, 3, 2, false, NT, 8> = MT 1, T 3, U 3, SwiGLU epilogue, bf16 weights, predicate-free form on 8 waves (RMSNorm-scaled w1|w3 GEMM; NT = non-temporal weight loads: the 10 slow layers' launches, not the 32 depth launches)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_kind": "bytes the L2s requested from the fabric (L2 misses; Infinity-Cache hits are counted), not HBM bytes: of this kernel's 42 launches per "
                                    "frame the 10 of the slow blocks stream their weights from HBM (non-temporal loads), the 32 of the depth transformer re-read "
                                    "weights the Infinity Cache holds -- `peak` is the HBM rate and a loose ceiling for those", "avg_us": round(avg_us, 3), "rocprof_avg_us": rocprof_us, "rocprof_source": rocprof_src,
                    "launches_timed": 8 * n_per_frame * 3, "bytes_per_launch": bytes_alg,
                    "method": "in situ: frame graph replayed with this kernel's launches duplicated (smoltts_session_measure_duplicate), HIP events on the launch stream; frac = achieved / peak from avg_us",
                    "step": {"bytes_per_frame_step": int(sb), "parts": {k: int(v) for k, v in parts.items()}, "mean_context": round(L_mean, 1),
                             "us_per_frame_step": round(us_per_frame_step, 1), "achieved": round(step_ach, 1), "unit": "GB/s",
                             "frac": round(step_ach / HBM_PEAK_GBS, 4),
                             "note": "SURVEY.md §8d algorithmic bytes of one frame for all B utterances (fast weights re-streamed for each of the 8 depth steps, KV at bf16 size) over the measured time per frame-step"}}

    # ---- CPU baseline: the oracle (fp32 torch eager) on the same weights/prompts, bounded sample; its ids and PCM are
    #      also what the GPU's first frames are compared with.  Then every timed frame is checked through the oracle's
    #      teacher-forced logits on the slot that saw the run's smallest top-2 gap.
    cpu = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        from oracle.lm_oracle import LMOracle, OracleLMConfig
        from oracle.mimi_oracle import MimiDecodeOracle

        nF = min(args.cpu_frames, n_total - 1)
        log(f"CPU oracle sample: prefill {B} prompts + {nF + 1} frames on {torch.get_num_threads()} threads")
        ocfg, ostate = (cfg, state)
        if args.weights == "fp8":  # the engine computes the dequantised model: that is what the oracle gets
            from smoltts_amd.packing import fp8_reference_state

            ocfg, ostate = fp8_reference_state(cfg, state)
        okw = dict(embed_mask=numerics.embed_mask, rope_bf16=numerics.rope_bf16, kv_bf16=(args.kv == "bf16"))
        orc = LMOracle(OracleLMConfig.from_dict(ocfg.__dict__), ostate, **okw)
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
                if f % 8 == 0:
                    log(f"  oracle frame {f} done")
            grid = torch.stack(cols_all, dim=1)  # B, nF+1, 9
            morc = MimiDecodeOracle(mstate)
            t1 = time.perf_counter()
            ref_pcm = morc.decode(grid[:, 1:, 1:].permute(0, 2, 1).contiguous())
            t_mimi = time.perf_counter() - t1
            log(f"  oracle Mimi decode done ({t_mimi:.1f}s)")
            del ref_pcm
            # BASELINE configs[0] / SURVEY.md §8d config 1: smoltts_byte_70m, one utterance, the same loop
            cfg70 = named_config("smoltts_byte_70m")
            st70 = synthetic_lm_state(cfg70, seed=0)
            o70 = LMOracle(OracleLMConfig.from_dict(cfg70.__dict__), st70, embed_mask=numerics.embed_mask, rope_bf16=numerics.rope_bf16)
            p70 = torch.from_numpy(all_prompts[0]).long()
            o70._alloc(1, p70.shape[1] + nF + 2)
            h70 = o70.prefill_one(0, p70)[None]
            t70, c70 = 0.0, []
            for f in range(nF + 1):
                t1 = time.perf_counter()
                ids = o70.slow_head(h70).argmax(-1)
                cds, _ = o70.fast_decode(h70)
                cols = torch.cat([ids[:, None], cds], dim=1)
                h70 = o70.decode_cols(cols)
                if f > 0:
                    t70 += time.perf_counter() - t1
                c70.append(cols)
            g70 = torch.stack(c70, dim=1)
            t1 = time.perf_counter()
            morc.decode(g70[:, 1:, 1:].permute(0, 2, 1).contiguous())
            t70m = time.perf_counter() - t1
        cpu = {"value": round(B * nF / (t_lm + t_mimi), 2), "unit": "frames/s", "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(),
               "kind": "port", "sample": f"{nF} decode frames x {B} utterances ({args.model} oracle, fp32 torch eager, KV-cached) "
               f"+ Mimi decode of those frames; LM {t_lm:.2f}s, Mimi {t_mimi:.2f}s; prefill and 1 warm-up frame untimed",
               "cores_note": "threads = the CPUs this process may run on, capped at the 16-CPU share of a one-GPU box (gpurun); a sample with "
                             "torch.set_num_threads(os.cpu_count() = 256) was tried in round 4 and did not finish one frame in 7 minutes under that share "
                             "(256 OpenMP threads on 16 CPUs), so there is no all-CPU figure",
               "b1_70m": {"value": round(nF / (t70 + t70m), 2), "unit": "frames/s", "sample": f"{nF} decode frames x 1 utterance (smoltts_byte_70m oracle) + Mimi decode; "
                          f"LM {t70:.2f}s, Mimi {t70m:.2f}s (BASELINE configs[0])"}}
        same = np.array_equal(codes[:, : nF + 1], grid.numpy())
        pcm_ref_full = morc.decode(grid[:, :, 1:].permute(0, 2, 1).contiguous())[:, 0].numpy()
        rms = float(np.sqrt(np.mean((pcm[:, : (nF + 1) * 1920].cpu().numpy() - pcm_ref_full) ** 2))) if not args.no_mimi else None
        # every frame of the run (warm-up and timed steps), teacher-forced, on the slot with the smallest top-2 gap
        b_min = int(np.argmin(margin))
        f_min, s_min = int(margin_at[b_min]) // 64, int(margin_at[b_min]) % 64
        log(f"smallest top-2 gap of the run: {margin[b_min]:.3e} at slot {b_min}, frame {f_min}, step {s_min} (0 = slow id); "
            f"teacher-forced oracle check of all {n_total} frames of that slot")
        with torch.no_grad():
            gb = codes[b_min, :n_total].T  # (9, F)
            full = torch.cat([torch.from_numpy(mine[b_min]).long(), torch.from_numpy(gb.astype(np.int64))], dim=1)
            tl, cl = orc.teacher_forced(full)
        Tb = mine[b_min].shape[1]
        flips, worst, gap_there = 0, 0.0, None
        for f in range(n_total):
            s_ = Tb - 1 + f
            for i, lg in enumerate([tl[s_]] + [cl[s_, k] for k in range(cl.shape[1])]):
                want, got = int(lg.argmax()), int(gb[i, f])
                if f == f_min and i == s_min:
                    top2 = torch.topk(lg, 2).values
                    gap_there = float(top2[0] - top2[1])
                if want != got:
                    flips += 1
                    worst = max(worst, float(lg[want] - lg[got]) / float(lg.abs().max()))
        parity = {"frames_checked": nF + 1, "ids_bit_identical": bool(same), "pcm_rms_err": rms,
                  "all_frames_teacher_forced": {"slot": b_min, "frames": n_total, "ids": n_total * (1 + cfg.max_fast_seqlen), "ids_differing_from_oracle_argmax": flips,
                                                "largest_relative_gap_at_a_difference": worst,
                                                "min_margin": {"engine_gap": float(margin[b_min]), "slot": b_min, "frame": f_min, "step": s_min,
                                                               "oracle_gap_there": gap_there,
                                                               "timed_step": (f_min - 1) // CH - W if f_min > W * CH else None}}}
        log(f"  teacher-forced: {flips} of {n_total * (1 + cfg.max_fast_seqlen)} ids differ from the oracle's argmax; oracle gap at the engine's minimum: {gap_there}")

    # ---- first-audio-chunk latency (second half of BASELINE.json's metric): submit -> first 1920 PCM
    #      samples on the host = prompt prefill + frame 0 + one Mimi step.  (a) this workload: all B
    #      utterances submitted together; (b) BASELINE configs[1]: smoltts_byte_70m, one utterance.
    first_chunk = None
    if rank == 0 and world == 1 and not args.no_latency:
        def first_chunk_ms(engine, mengine, prompts, reps, kv="fp32"):
            ts = []
            ls = LMSession(engine, max_batch=len(prompts), max_seq=max(p.shape[1] for p in prompts) + 8,
                           max_rows=sum(p.shape[1] for p in prompts), max_frames=4, kv_dtype=kv)
            ms = MimiSession(mengine, max_batch=len(prompts), max_chunk_frames=1)
            buf = torch.zeros(len(prompts), 1920, dtype=torch.float32, device=dev)
            for _ in range(reps + 1):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                do_prefill(ls, prompts, stop_on_eos=False)
                ms.reset()
                ms.decode_chunk(ls.codes, 0, 1, buf, code_offset=1)
                _ = buf.cpu()
                ts.append((time.perf_counter() - t1) * 1e3)
            ms.close(); ls.close()
            return ts[1:]  # first repetition warms the allocator

        t150 = first_chunk_ms(eng, meng, mine, 5, args.kv)

        def steady_first_chunk_ms(engine, mengine, prompts, reps):
            """Continuous batching: all B slots are speaking; one slot is restarted with a new prompt (prefill of that
            prompt + frame 0) and its first frame goes through a one-slot Mimi session to the host."""
            ls = LMSession(engine, max_batch=len(prompts), max_seq=max(p.shape[1] for p in prompts) + 64,
                           max_rows=sum(p.shape[1] for p in prompts), max_frames=64, kv_dtype=args.kv)
            ls.prefill(prompts, stop_on_eos=False)
            ls.decode(4)
            ms1 = MimiSession(mengine, max_batch=1, max_chunk_frames=1)
            buf = torch.zeros(1, 1920, dtype=torch.float32, device=dev)
            ts = []
            for i in range(reps + 2):
                k = i % len(prompts)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                do_prefill(ls, [prompts[(k + 7) % len(prompts)]], slots=[k], stop_on_eos=False)
                ms1.reset()
                ms1.decode_chunk(ls.codes[k:k + 1], 0, 1, buf, code_offset=1)
                _ = buf.cpu()
                ts.append((time.perf_counter() - t1) * 1e3)
                ls.decode(1)  # the other slots keep speaking between arrivals
            ms1.close(); ls.close()
            return ts[2:]

        t150s = steady_first_chunk_ms(eng, meng, mine, 30)
        cfg70 = named_config("smoltts_byte_70m")
        a70, o70_ = pack_lm(cfg70, synthetic_lm_state(cfg70, seed=0), numerics)
        eng70 = LMEngine(cfg70, None, TokenConfig.from_tokenizer(tok, cfg70), numerics, arena=a70, offsets=o70_)
        t70l = []
        for u in range(200):  # p50 / p95 over 200 prompts
            t70l += first_chunk_ms(eng70, meng, [all_prompts[u % len(all_prompts)]], 1)
        # BASELINE configs[1]: one 70m stream, every frame decoded to PCM and copied to the host as it appears -- the façade's
        # streaming loop (generate.stream_pcm: the codec step of frame f beside frame f + 1), and the one-stream loop beside it
        from smoltts_amd.generate import stream_pcm

        def b1_stream(overlap):
            ls1 = LMSession(eng70, max_batch=1, max_seq=400, max_rows=256, max_frames=150)
            ms1 = MimiSession(meng, max_batch=1, max_chunk_frames=1)
            t1, n = None, 0
            for f, chunk in enumerate(stream_pcm(ls1, ms1, all_prompts[0], stop_on_eos=False, overlap=overlap)):
                if f == 21:  # the first frames warm the graph and the allocator
                    t1 = time.perf_counter()
                n = f
            fps = (n - 21) / (time.perf_counter() - t1)
            ms1.close(); ls1.close()
            return fps

        b1_stream_fps, b1_serial_fps = b1_stream(True), b1_stream(False)
        first_chunk = {"b1_70m_stream_frames_per_s": round(b1_stream_fps, 1), "b1_70m_stream_frames_per_s_one_stream": round(b1_serial_fps, 1), "batch": B, "model": args.model,
                       "batch_at_once_ms_p50": round(float(np.median(t150)), 2),
                       "steady_one_arrival_ms_p50": round(float(np.median(t150s)), 2), "steady_one_arrival_ms_p95": round(float(np.percentile(t150s, 95)), 2),
                       "b1_70m_ms_p50": round(float(np.median(t70l)), 2), "b1_70m_ms_p95": round(float(np.percentile(t70l, 95)), 2),
                       "b1_70m_prompts": len(t70l), "prefill_chunk": pch or None,
                       "includes": f"prefill + frame 0 + Mimi step + D2H copy of 1920 samples; batch_at_once: all {B} prompts of this run's model submitted together; "
                                   f"steady_one_arrival: one new prompt into a session whose {B} slots are all speaking"
                                   + (f"; prompts enter in chunks of {pch} columns (chunked prefill inside the timed figure)" if pch else "")}
        log(f"first audio chunk: B={B} {args.model} p50 {first_chunk['batch_at_once_ms_p50']} ms (all at once) / {first_chunk['steady_one_arrival_ms_p50']} ms "
            f"(one arrival among {B} speaking slots); B=1 70m p50 {first_chunk['b1_70m_ms_p50']} ms")

    if rank == 0:
        wdesc = ("fp8-e4m3 weights (storage format, row-scaled; dequantised to bf16 in registers, so the MFMA operands are bf16 -- no fp8 MFMA)"
                 if args.weights == "fp8" else "bf16 weights")
        out = {
            "metric": "Mimi frames/sec (=12.5 x RTF), greedy DualAR decode + Mimi decode to PCM" + (" [DIAGNOSTIC: Mimi skipped]" if args.no_mimi else "")
                      + (" [fp8 MFMA prefill: prompt KV rows approximate, ids NOT comparable with the reference greedy decode]" if args.fp8_prefill else ""),
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "ranks_seen": seen, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": f"{wdesc}, fp32 activations/accumulate, {args.kv} KV cache (LM); " + ("fp32 (Mimi)" if args.codec_products == 6 else "Mimi on hi + mid bf16 operand pieces, three exact products per pair, fp32 accumulate (--codec-products 3: 2^-16-grade, not the default)"),
            "data": "synthetic (seeded random weights at the real shapes, synthetic ChatML prompts)",
            "config": {"workload": f"{args.model} B={B}/GPU concurrent utterances, chunk {CH} frames/step, "
                                   f"prompts T={min(p.shape[1] for p in mine)}..{max_T}, context {max_T + W * CH}..{max_T + (W + K) * CH}"
                                   + (f", chunked prefill ({pch} columns)" if pch else ""),
                       "global_batch": B * world, "frames_per_step": B * CH * world, "parallelism": f"dp{world} (utterance-sharded replicas)", "streams_per_gpu": S,
                       "codec": "no Mimi decode (--no-mimi)" if args.no_mimi else (
                           f"chunk i's Mimi decode on a second stream beside chunk i + 1's frame graphs ({args.overlap_wait}-side hand-over"
                           + (f", {mimi_cus_used}-CU mask" if mimi_cus_used else "") + ")" if args.overlap_mimi else "Mimi decode on the frame graphs' stream, after them")},
            "rtf": round(value / 12.5, 1), "frames_per_s_per_gpu": round(value / world, 1),
            "per_rank_frames_per_s": {"min": round(min(rank_fps), 1), "max": round(max(rank_fps), 1), "ranks": len(rank_fps)},
            "us_per_frame_step": round(us_per_frame_step, 1), "prefill_ms": round(prefill_ms, 2),
            "min_top2_margin": float(margin.min()),
            "first_audio_chunk": first_chunk, "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
            "launcher_fallback": launcher_record()[0], "pinned_attempt_failed": launcher_record()[1],
        }
        out["parity"] = dict(parity or {}, **rank_parity)
        print(json.dumps(out), flush=True)
    for x in msessions + sessions:
        x.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    if ranks_passed != ranks_checked:  # a rank whose GPU does not reproduce the golden grid: the line above says so, the run fails
        raise SystemExit(f"rank {rank}: {ranks_checked - ranks_passed} rank(s) failed the golden-grid check")


def main():
    args = parse_args()
    if args.fp8_prefill:
        if args.weights != "fp8":
            raise SystemExit("--fp8-prefill needs --weights fp8")
        os.environ["SMOLTTS_FP8_PREFILL"] = "1"  # every LMSession of this process (children inherit it)
        args.cpu_frames = 0  # no id parity with the oracle in this mode
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # not under a launcher: become one (before any GPU call)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.rehearse_launcher:
        return rehearse(args)
    run_rank(args)


if __name__ == "__main__":
    main()
