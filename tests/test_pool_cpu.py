"""The multi-GPU serving front-end (smoltts_amd/server/pool.py) without GPUs: worker processes run a stand-in scheduler,
so what is tested is the routing, the relay, cancellation and failure handling — the part that is the same on 8 MI355X."""
import functools
import threading
import time

import numpy as np
import pytest

from pool_helpers import make_broken, make_echo


@pytest.fixture(autouse=True)
def _no_inherited_mask(monkeypatch):
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)  # a GPU box may run the suite under a one-device mask


def test_visible_device_respects_an_inherited_mask():
    from smoltts_amd.server.pool import visible_device

    assert visible_device(3, None) == "3"
    assert visible_device(1, "4,6,7") == "6"
    with pytest.raises(ValueError):
        visible_device(3, "4,6,7")


def test_pool_routes_relays_and_balances():
    from smoltts_amd.server.pool import GpuPool

    pool = GpuPool(functools.partial(make_echo, 0.01), devices=[0, 1], ready_timeout=120)
    try:
        texts = ["a" * n for n in (5, 9, 3, 7, 4, 6)]
        got = [None] * len(texts)

        def client(i):
            r = pool.submit(texts[i], "sky", stream=bool(i % 2))
            got[i] = np.stack(list(pool.iter_chunks(r)))

        threads = [threading.Thread(target=client, args=(i,)) for i in range(len(texts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
        served = []
        for t, g in zip(texts, got):
            assert g is not None and g.shape == (len(t), 4)
            assert (g[:, 1] == np.arange(len(t))).all() and (g[:, 2] == 3).all() and (g[:, 3] == len(t)).all()  # in order, own request
            assert len(set(g[:, 0])) == 1  # one worker served the whole request
            served.append(int(g[0, 0]))
        assert sorted(set(served)) == [0, 1]  # both workers were used, each under its own device mask
        assert pool.loads() == [0, 0] and pool.stats() == {"workers": 2, "alive": 2, "in_flight": [0, 0], "restarts": [0, 0]}
        assert pool.synthesize("abcd", max_new_tokens=2).shape == (8,)
        # an error raised for one request reaches that client only
        with pytest.raises(ValueError, match="bad request"):  # the type survives the process boundary (HTTP 400, not 500)
            pool.synthesize("__raise__")
        assert pool.synthesize("ok").shape == (8,)
    finally:
        pool.close()
    with pytest.raises(RuntimeError):
        pool.submit("after close")


def test_pool_cancel_and_worker_death():
    from smoltts_amd.server.pool import GpuPool

    pool = GpuPool(functools.partial(make_echo, 0.02), devices=[0, 1], ready_timeout=120, respawn=False)
    try:
        r = pool.submit("x" * 500, stream=True)  # 10 s of chunks if nobody stops it
        it = pool.iter_chunks(r)
        next(it)
        it.close()  # the HTTP client went away
        assert r.cancelled
        deadline = time.time() + 20
        while pool.loads()[r.worker] and time.time() < deadline:
            time.sleep(0.05)
        assert pool.loads() == [0, 0]  # the worker ended the request long before its 500 chunks
        # the scheduler of that worker saw exactly one cancel (ask until the request lands on it)
        seen = 0
        for _ in range(4):
            q = pool.submit("__cancels__")
            val = list(pool.iter_chunks(q))
            if q.worker == r.worker:
                seen = int(val[0][0])
                break
            # occupy the other worker so that the next question goes to r.worker
            hold = pool.submit("y" * 50)
            q2 = pool.submit("__cancels__")
            val = list(pool.iter_chunks(q2))
            list(pool.iter_chunks(hold))
            if q2.worker == r.worker:
                seen = int(val[0][0])
                break
        assert seen == 1
        # a worker that dies takes its requests down with an error, the other one keeps serving
        victim = pool.submit("z" * 400, stream=True)
        killer = None
        for _ in range(3):  # the killer must land on the victim's worker: load the other one first
            other = pool.submit("w" * 100)
            killer = pool.submit("__die__")
            if killer.worker == victim.worker:
                break
            killer = None
        assert killer is not None
        with pytest.raises(RuntimeError, match="died"):
            list(pool.iter_chunks(victim))
        with pytest.raises(RuntimeError, match="died"):
            list(pool.iter_chunks(killer))
        assert np.stack(list(pool.iter_chunks(other))).shape == (100, 4)
        g = np.stack(list(pool.iter_chunks(pool.submit("abc"))))
        assert g.shape == (3, 4) and int(g[0, 0]) != int(victim.worker)
    finally:
        pool.close()


def test_pool_replaces_a_worker_that_died():
    from smoltts_amd.server.pool import GpuPool

    pool = GpuPool(functools.partial(make_echo, 0.0), devices=[0, 1], ready_timeout=120)
    try:
        killer = pool.submit("__die__")
        with pytest.raises(RuntimeError, match="died"):
            list(pool.iter_chunks(killer))
        deadline = time.time() + 60
        while pool.stats()["alive"] < 2 and time.time() < deadline:
            time.sleep(0.1)
        st = pool.stats()
        assert st["alive"] == 2 and st["restarts"][killer.worker] == 1
        served = set()
        for i in range(8):  # both workers serve again, the replacement under the dead one's device mask
            hold = pool.submit("h" * 30)  # goes to the first idle worker, the next request to the other one
            g = np.stack(list(pool.iter_chunks(pool.submit("abc"))))
            h = np.stack(list(pool.iter_chunks(hold)))
            served |= {int(g[0, 0]), int(h[0, 0])}
        assert served == {0, 1}
    finally:
        pool.close()


def test_pool_reports_a_worker_that_cannot_start():
    from smoltts_amd.server.pool import GpuPool

    with pytest.raises(RuntimeError, match="no checkpoint here"):
        GpuPool(make_broken, devices=[0], ready_timeout=120)
