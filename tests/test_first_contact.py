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


def _start(rank, world, port, *args):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
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
