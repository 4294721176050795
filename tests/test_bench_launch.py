"""`python bench.py --gpus N` must start N ranks by itself (the driver's SCALE run calls it exactly like that), give
every rank its own contiguous realization range and reduce the integer counters once.  --launch-check runs that
machinery on gloo with an integer checksum per realization index in place of the kernel, so it runs without a GPU:
two ranks must cover the same indices as one rank with twice the steps and arrive at the same counters."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--launch-check"] + list(argv), env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


@pytest.mark.timeout(600)
def test_gpus_flag_launches_that_many_ranks_with_disjoint_ranges():
    one = _bench("--gpus", "1", "--steps", "6", "--batch", "512")
    two = _bench("--gpus", "2", "--steps", "3", "--batch", "512")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["rank_ranges"] == [[0, 3072]]
    assert two["rank_ranges"] == [[0, 1536], [1536, 3072]]          # disjoint, contiguous, same union
    assert two["counters"] == one["counters"]
    assert one["counters"]["n_realizations"] == 3072 and one["counters"]["sym_errors"] > 0


@pytest.mark.timeout(600)
def test_strong_split_covers_the_literal_total_once():
    """The strong-scaling leg (BASELINE config 4: 10^6 realizations over the ranks; here 10 007 to keep the remainder
    visible): contiguous blocks floor(T r / N) .. floor(T (r + 1) / N), union = [0, T), reduced counters = the one-rank ones."""
    one = _bench("--gpus", "1", "--steps", "2", "--batch", "512", "--strong-total", "10007")
    two = _bench("--gpus", "2", "--steps", "1", "--batch", "512", "--strong-total", "10007")
    assert one["strong"]["rank_ranges"] == [[0, 10007]]
    assert two["strong"]["rank_ranges"] == [[0, 5003], [5003, 10007]]
    assert two["strong"]["counters"] == one["strong"]["counters"]
    assert one["strong"]["counters"]["n_realizations"] == 10007 and one["strong"]["total_realizations"] == 10007
    default = _bench("--gpus", "2", "--steps", "1", "--batch", "65536")
    assert default["strong"]["total_realizations"] == 10 ** 6           # config 4's literal count
    assert default["strong"]["rank_ranges"] == [[0, 500000], [500000, 1000000]]


def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--launch-check", "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
