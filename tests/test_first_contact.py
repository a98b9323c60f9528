"""First contact of `bench.py --gpus N` is bounded and says where it stopped (the driver launches one rank per GPU
with torch.distributed.run and times the whole run from outside).  A rank that never joins must not leave the others waiting for
the default 600 s: the rendezvous carries --init-timeout, a watchdog thread names the phase on stderr and ends the process with
exit code 3 after its budget.  CPU tests: the gloo rendezvous happens before anything touches a GPU."""
import os
import socket
import subprocess
import sys
import time

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _start(rank, world, port, *args, **extra_env):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.update(extra_env)
    return subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", "2",
                             "--warmup", "1"] + [str(a) for a in args], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_a_rank_that_never_joins_ends_the_others_inside_the_timeout():
    # world 3, rank 2 is never started: ranks 0 and 1 must both give up within --init-timeout (+ start-up), non-zero, naming the phase
    port = _free_port()
    t0 = time.time()
    procs = [_start(r, 3, port, "--init-timeout", 8) for r in (0, 1)]
    outs = [p.communicate(timeout=120) for p in procs]
    took = time.time() - t0
    assert took < 60, took
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 4, (p.returncode, err[-1500:])
        assert "phase: rendezvous (gloo init_process_group, world 3" in err and "rendezvous failed" in err and "ranks arrived" in err, err[-1500:]
        assert not [l for l in out.splitlines() if l.startswith("{")]          # no JSON line from a run that never started


def test_the_watchdog_ends_a_stuck_run_with_exit_code_3_and_names_the_phase():
    # the rendezvous may wait 60 s here, the watchdog's budget is 4 s: it must end the process first, with the phase in its message
    port = _free_port()
    t0 = time.time()
    p = _start(0, 2, port, "--init-timeout", 60, "--watchdog-seconds", 4)
    out, err = p.communicate(timeout=120)
    assert time.time() - t0 < 45
    assert p.returncode == 3, (p.returncode, err[-1500:])
    assert "WATCHDOG" in err and "in phase 'rendezvous (gloo init_process_group, world 2" in err and "exit code 3" in err, err[-1500:]


# ---- the link check (bench_sharded.link_check): one message from every peer into rank 0 before anything depends on a link ----------
def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_the_link_check_prints_a_table_and_lets_a_healthy_run_go_on():
    # world 2 through gloo (host buffers): the check passes, rank 0 prints one row per peer; without a GPU the run then ends where the
    # product says it has no CPU path -- after the check, not in it
    port = _free_port()
    procs = [_start(r, 2, port, "--init-timeout", 30, "--watchdog-seconds", 60) for r in (0, 1)]
    outs = [p.communicate(timeout=240) for p in procs]
    err0 = outs[0][1]
    assert "link check (one message from every peer into rank 0, every byte compared)" in err0, err0[-2000:]
    assert "rank 1 (host buffers) -> rank 0 (host buffers)" in err0 and ": ok" in err0, err0[-2000:]
    if _no_gpu():
        for p, (out, err) in zip(procs, outs):
            assert p.returncode not in (0, 3, 4, 5) and "no CPU path" in err, (p.returncode, err[-1500:])


def test_a_link_that_delivers_other_bytes_ends_every_rank_with_the_link_named():
    # rank 1's payload is damaged on the way (test hook SDFHIP_BENCH_LINK_FAULT): rank 0 finds the bytes that differ, tells everyone
    # in the verdict's broadcast, and ALL THREE ranks end with exit code 5 naming the link -- also rank 2, whose own link is fine
    port = _free_port()
    t0 = time.time()
    procs = [_start(r, 3, port, "--init-timeout", 30, "--watchdog-seconds", 60, SDFHIP_BENCH_LINK_FAULT="corrupt:1") for r in (0, 1, 2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert time.time() - t0 < 120
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 5, (r, p.returncode, err[-1500:])
        assert "link check FAILED: rank 1 (host buffers) -> rank 0 (host buffers) delivered other bytes" in err, (r, err[-1500:])
        assert not [l for l in out.splitlines() if l.startswith("{")]
    assert "1 of 1048576 bytes differ from what rank 1 sent" in outs[0][1]
    assert "rank 2 (host buffers) -> rank 0 (host buffers)" in outs[0][1]          # the table still holds the healthy link's row


def test_a_silent_link_ends_rank_0_inside_the_limit_and_the_peer_non_zero():
    # rank 1 never sends: rank 0 waits --link-timeout seconds, prints the table with that row and leaves with exit code 5; rank 1 is
    # left in the verdict's broadcast, which fails once rank 0 is gone (or its watchdog ends it): non-zero either way
    port = _free_port()
    t0 = time.time()
    procs = [_start(r, 2, port, "--init-timeout", 30, "--link-timeout", 3, "--watchdog-seconds", 25, SDFHIP_BENCH_LINK_FAULT="silent:1")
             for r in (0, 1)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert time.time() - t0 < 120
    assert procs[0].returncode == 5, (procs[0].returncode, outs[0][1][-1500:])
    assert "NOTHING ARRIVED within 3 s" in outs[0][1] and "link check FAILED: rank 1 (host buffers) -> rank 0 (host buffers)" in outs[0][1]
    assert procs[1].returncode != 0, outs[1][1][-1500:]


def test_the_nccl_branch_of_the_link_check_bounds_a_receive_by_an_event():
    # No two GPUs here: the NCCL branch of bench_sharded.link_check is exercised against stand-ins for torch / torch.distributed.
    # The receive is bounded by QUERYING an event recorded behind it (a wait with a timeout does not bound a receive on the GPU): a
    # link whose event completes passes; one whose event never completes ends the process with exit code 5 inside the limit.
    import subprocess
    prog = r"""
import sys, time, types
sys.path.insert(0, %r)
import torch as real_torch
import bench_sharded as bs

class Work:
    def wait(self, *a): pass
    def is_completed(self): raise AssertionError("the NCCL branch must not rely on is_completed()")
class Dist:
    sent = []
    @staticmethod
    def irecv(buf, src):
        i = real_torch.arange(buf.numel(), dtype=real_torch.int64)
        buf.copy_(((i * 131 + src * 17 + (i >> 9)) %% 251).to(real_torch.uint8))       # what rank `src` sends
        return Work()
    @staticmethod
    def broadcast(t, src): pass
class Event:
    dead = %s
    def __init__(self): self.t = None
    def record(self): self.t = time.perf_counter()
    def query(self): return (not Event.dead) and time.perf_counter() - self.t > 0.05     # the bytes arrive 50 ms after the post
class Cuda:
    Event = Event
    @staticmethod
    def synchronize(): pass
class Torch:                       # torch with "cuda" tensors living on the CPU
    cuda = Cuda
    int64, int32, uint8 = real_torch.int64, real_torch.int32, real_torch.uint8
    @staticmethod
    def arange(n, dtype=None, device=None): return real_torch.arange(n, dtype=dtype)
    @staticmethod
    def zeros(*a, dtype=None, device=None): return real_torch.zeros(*a, dtype=dtype)
    @staticmethod
    def equal(a, b): return real_torch.equal(a, b)
class Wd:
    def phase(self, *a, **k): pass
rows = bs.link_check(Torch, Dist, 0, 3, True, Wd(), ["0000:05:00.0", "0000:15:00.0", "0000:25:00.0"], limit=1.0, nbytes=1 << 16, peer_access=[True, True, False])
print("ROWS", [(r["verdict"], r["peer_access"]) for r in rows])
""" 
    ok = subprocess.run([sys.executable, "-c", prog % (REPO, "False")], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and "ROWS [('ok', True), ('ok', False)]" in ok.stdout, ok.stdout + ok.stderr
    assert "peer access NO (copies stage through the host)" in ok.stderr and "rank 2 (0000:25:00.0) -> rank 0 (0000:05:00.0)" in ok.stderr
    t0 = time.time()
    dead = subprocess.run([sys.executable, "-c", prog % (REPO, "True")], capture_output=True, text=True, timeout=120)
    assert dead.returncode == 5 and time.time() - t0 < 30, (dead.returncode, dead.stderr[-800:])
    assert "NOTHING ARRIVED within 1 s" in dead.stderr and "link check FAILED: rank 1 (0000:15:00.0) -> rank 0 (0000:05:00.0)" in dead.stderr
