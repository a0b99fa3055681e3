"""N > 1 path on CPU: two gloo ranks exercise the utterance sharding, the weight broadcast and the
max/sum reductions exactly as bench.py uses them under RCCL."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from smoltts_amd import parallel
    from smoltts_amd.config import NumericsMode
    from smoltts_amd.packing import pack_lm
    from smoltts_amd.synthetic import named_config, synthetic_lm_state

    r, w, _ = parallel.init_distributed(backend="gloo")
    cfg = named_config("tiny")
    arena = offsets = None
    if r == 0:
        arena, offsets = pack_lm(cfg, synthetic_lm_state(cfg, seed=0), NumericsMode.torch_reference())
    arena, offsets = parallel.broadcast_weights(arena, offsets, torch.device("cpu"))
    mine = parallel.shard_utterances(7, r, w)
    t = parallel.all_reduce_max(1.0 + r, "cpu")
    n = parallel.all_reduce_sum(float(len(mine)), "cpu")
    parallel.barrier()
    q.put((r, int(arena.numel()), int(arena.to(torch.int64).sum()), offsets["layers"][1]["w2"], mine, t, n))
    torch.distributed.destroy_process_group()


def test_two_rank_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, s0, o0, m0, t0, c0), (r1, n1, s1, o1, m1, t1, c1) = out
    assert (n0, s0, o0) == (n1, s1, o1) and n0 > 0          # identical arena bytes + offsets on both ranks
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]            # u -> rank u mod world, no overlap, full cover
    assert t0 == t1 == 2.0 and c0 == c1 == 7.0               # max over ranks, sum over ranks


def test_single_process_helpers_are_noops():
    from smoltts_amd import parallel

    assert parallel.shard_utterances(5, 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.all_reduce_max(3.5, "cpu") == 3.5 and parallel.all_reduce_sum(2.0, "cpu") == 2.0
    a, o = parallel.broadcast_weights(torch.arange(4, dtype=torch.uint8), {"x": 1}, torch.device("cpu"))
    assert a.tolist() == [0, 1, 2, 3] and o == {"x": 1}
