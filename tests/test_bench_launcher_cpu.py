"""`python bench.py --gpus N` must start its own ranks (the driver runs exactly that command): the parent spawns N children
before any GPU call, relays rank 0's JSON line and returns the children's exit code.  Rehearsed here on CPU over gloo
(`--rehearse-launcher`: the bench's distributed plumbing without compute)."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(extra_env=None, *args, timeout=240):
    env = dict(os.environ, TORCH_COMPILE_DISABLE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    return r, time.time() - t0


def test_bench_self_launches_two_ranks_and_reports_ranks_seen():
    r, _ = _run(None, "--gpus", "2", "--rehearse-launcher", "--steps", "2", "--warmup", "1", "--batch", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2                 # the all-reduce of ones saw both ranks
    assert d["utterances_sharded"] == 6 and d["arena_identical_on_all_ranks"] is True and d["arena_bytes"] > 0
    assert d["value"] is None and "REHEARSAL" in d["metric"]         # never mistaken for a measurement
    # the per-rank self-check plumbing of the N > 1 run: arena checksums all-gathered and compared on every rank, every
    # rank's own check summed over the backend (in the real run: the golden grid decoded on each rank's GPU)
    assert d["parity"] == {"ranks_checked": 2, "ranks_passed": 2, "arena_checksums_identical": True}


def test_a_failing_rank_ends_the_run_with_its_exit_code():
    """Rank 1 dies after the rendezvous; rank 0 is then stuck in a collective: the launcher must stop it and return
    non-zero instead of hanging."""
    r, took = _run({"SMOLTTS_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--rehearse-launcher", timeout=240)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 3" in r.stderr and took < 200
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_a_start_up_failure_under_device_masks_is_retried_once_without_them():
    """If the pinned attempt dies before rank 0 has printed anything, the launcher tries once more with the binding left to
    LOCAL_RANK (the torch.distributed.run way) and says so."""
    r, _ = _run({"SMOLTTS_BENCH_FAIL_IF_PINNED": "1"}, "--gpus", "2", "--rehearse-launcher", "--batch", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "retrying ONCE without per-rank HIP_VISIBLE_DEVICES masks" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["ranks_seen"] == 2
    d = json.loads(lines[0])  # the record says that the line comes from the second attempt, and what became of the first
    assert d["launcher_fallback"] is True and d["pinned_attempt_failed"]["rc"] == 5 and d["pinned_attempt_failed"]["rank"] in (0, 1)
    # ... and with the fallback switched off the failure stands
    r, _ = _run({"SMOLTTS_BENCH_FAIL_IF_PINNED": "1", "SMOLTTS_BENCH_NO_FALLBACK": "1"}, "--gpus", "2", "--rehearse-launcher")
    assert r.returncode == 5


def test_a_rank_killed_by_a_signal_is_not_retried():
    """A rank that dies of a signal at start-up (a GPU memory fault shows as SIGSEGV / SIGABRT) must not be re-run on the box: no
    second attempt, exit code 128 + signal, nothing on stdout."""
    r, _ = _run({"SMOLTTS_BENCH_FAIL_IF_PINNED": "abort"}, "--gpus", "2", "--rehearse-launcher", "--batch", "3")
    assert r.returncode == 128 + 6, (r.returncode, r.stderr[-2000:])
    assert "retrying ONCE" not in r.stderr and "killed by signal 6: no retry" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_an_unretried_run_says_so_on_its_line():
    r, _ = _run(None, "--gpus", "2", "--rehearse-launcher", "--batch", "3")
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["launcher_fallback"] is False and d["pinned_attempt_failed"] is None


def test_under_an_external_launcher_the_process_is_a_rank():
    """torch.distributed.run sets WORLD_SIZE: bench.py must then NOT spawn anything (here: WORLD_SIZE=1 -> plain rank 0)."""
    r, _ = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "1", "--rehearse-launcher")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1
    r, _ = _run({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2", "--rehearse-launcher")
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr          # mismatch is an error, not a silent 1-rank run


def test_a_signal_to_the_launcher_takes_the_ranks_down():
    """SIGTERM to the parent (a driver's `timeout`): the ranks run in sessions of their own, so the parent must end them
    itself -- none may survive it holding a GPU."""
    import signal

    env = dict(os.environ, TORCH_COMPILE_DISABLE="1", SMOLTTS_BENCH_SLEEP_RANKS="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--rehearse-launcher"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 120
        kids = []
        while time.time() < deadline and len(kids) < 2:  # the children of the launcher, by parent PID
            time.sleep(0.5)
            out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(p.pid)], capture_output=True, text=True).stdout.split()
            kids = [int(x) for x in out]
        assert len(kids) == 2, kids
        time.sleep(1.0)
        p.send_signal(signal.SIGTERM)
        rc = p.wait(timeout=60)
        assert rc == 128 + signal.SIGTERM
        time.sleep(0.5)
        for k in kids:
            assert not Path(f"/proc/{k}").exists() or "Z" in Path(f"/proc/{k}/stat").read_text().split()[2], f"rank process {k} survived the launcher"
    finally:
        if p.poll() is None:
            p.kill()
