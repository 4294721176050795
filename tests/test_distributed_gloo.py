"""CPU, world_size 2 (gloo): the sharded batched loop gives bit-identical integer counters and
Result state for 1 and 2 ranks (realization-index sharding + one
all-reduce of the counter block).  The per-index 'errors' come from a deterministic fake so the
test needs no GPU; the GPU box exercises the same loop with the HIP pipelines."""
import json
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path, batch_size, early):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from test_simulations_cpu import FakeBatched
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if early:
            class Sim(FakeBatched):
                def _keep_going(self, p, res, rep):
                    return res["symbol_errors"][-1].get_result() < 20000
        else:
            Sim = FakeBatched
        sim = Sim(batch_size, rep_max=1500)
        sim.simulate()
        state = {n: [r.to_dict() for r in sim.results[n]] for n in ("ser", "ber", "symbol_errors", "bit_errors",
                                                                     "num_symbols", "num_bits", "num_skipped_reps")}
        with open("%s.%d" % (out_path, rank), "w") as fh:
            json.dump({"state": state, "reps": sim.runned_reps, "calls": sim.calls}, fh)
    finally:
        dist.destroy_process_group()


def _run(world, tmp_path, tag, batch_size, early=False):
    import torch.multiprocessing as mp
    out = str(tmp_path / tag)
    mp.spawn(_worker, args=(world, _free_port(), out, batch_size, early), nprocs=world, join=True)
    return [json.load(open("%s.%d" % (out, r))) for r in range(world)]


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank(tmp_path):
    one = _run(1, tmp_path, "w1", 128)[0]
    two = _run(2, tmp_path, "w2", 64)            # same global batch (64 per rank x 2)
    assert two[0]["state"] == two[1]["state"] == one["state"]
    assert two[0]["reps"] == one["reps"] == [1500, 1500]
    # the ranks really split the work: disjoint, contiguous index ranges covering the global batch
    a, b = two[0]["calls"], two[1]["calls"]
    assert len(a) == len(b) and all(x[0] + x[1] == y[0] for x, y in zip(a, b))
    assert sum(c for _, c in a) + sum(c for _, c in b) == sum(c for _, c in one["calls"])


@pytest.mark.timeout(300)
def test_two_ranks_early_stop_is_consistent(tmp_path):
    one = _run(1, tmp_path, "e1", 100, early=True)[0]
    two = _run(2, tmp_path, "e2", 50, early=True)
    assert two[0]["state"] == two[1]["state"] == one["state"]
    assert two[0]["reps"] == one["reps"] and all(r < 1500 for r in one["reps"])
