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
            json.dump({"state": state, "reps": sim.runned_reps, "calls": sim.calls, "n_reductions": sim.n_reductions},
                      fh)
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
    # ONE reduction per parameter variation and round of batches (a second / third round only replaces the
    # realizations the first one skipped) -- not one per batch: 2 variations x 24 batches here
    assert one["n_reductions"] == 0
    assert two[0]["n_reductions"] == two[1]["n_reductions"] <= 2 * 3 and len(a) >= 2 * 12


@pytest.mark.timeout(300)
def test_two_ranks_early_stop_is_consistent(tmp_path):
    one = _run(1, tmp_path, "e1", 100, early=True)[0]
    two = _run(2, tmp_path, "e2", 50, early=True)
    assert two[0]["state"] == two[1]["state"] == one["state"]
    assert two[0]["reps"] == one["reps"] and all(r < 1500 for r in one["reps"])
    # a stopping rule needs the global counters before every batch: one reduction per batch
    assert two[0]["n_reductions"] == len(two[0]["calls"])


def _worker_files(rank, world, port, folder, phase):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from test_simulations_cpu import FakeBatchedExtra
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sim = FakeBatchedExtra(50, rep_max=400 if phase == 0 else 1200)
        sim.partial_save_every_reps = 100
        sim.set_results_filename(os.path.join(folder, "shared"))
        sim.simulate()
        with open(os.path.join(folder, "out.%d.%d" % (phase, rank)), "w") as fh:
            json.dump({"cap": [r.to_dict() for r in sim.results["cap"]], "ser": [r.to_dict() for r in sim.results["ser"]],
                       "first_call": sim.calls[0], "reps": sim.runned_reps}, fh)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_files_resume_and_float_side_sums(tmp_path):
    """Rank-safe files: only rank 0 writes / removes, a resume point is read by rank 0 and broadcast;
    float side sums (the IA application's sum capacity) are all-reduced and survive the resume."""
    import torch.multiprocessing as mp
    from test_simulations_cpu import FakeBatchedExtra
    folder = str(tmp_path)
    mp.spawn(_worker_files, args=(2, _free_port(), folder, 0), nprocs=2, join=True)
    files = sorted(os.listdir(os.path.join(folder, "partial_results")))
    assert len(files) == 2 and os.path.exists(os.path.join(folder, "shared.pickle"))    # one per variation, once
    mp.spawn(_worker_files, args=(2, _free_port(), folder, 1), nprocs=2, join=True)
    outs = [json.load(open(os.path.join(folder, "out.1.%d" % r))) for r in range(2)]
    assert outs[0]["cap"] == outs[1]["cap"] and outs[0]["ser"] == outs[1]["ser"]
    assert outs[0]["reps"] == [1200, 1200]
    assert outs[0]["first_call"][0] >= 400                  # resumed past the first run's indices
    ref = FakeBatchedExtra(64, rep_max=1200)                # single rank, no files, one go
    ref.simulate()
    assert [r.to_dict() for r in ref.results["ser"]] == outs[0]["ser"]
    got, want = outs[0]["cap"], [r.to_dict() for r in ref.results["cap"]]
    for g, w in zip(got, want):
        assert g["num_updates"] == w["num_updates"] and abs(g["value"] - w["value"]) <= 1e-9 * abs(w["value"])


# ---- the product's own rendezvous (pyphysim_amd.distributed: the 128-byte RCCL id over TCP), eight ranks, no GPU --------------
class _StubEngine:
    """Stands in for Engine where NativeComm only needs the id source and mcle_comm_init: records who joined."""
    joined = []

    def __init__(self):
        self.inits = []

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, uid, rank, world):
        assert uid == bytes(range(128))
        self.inits.append((rank, world))
        _StubEngine.joined.append(rank)

    def new_counters(self):
        return None


@pytest.mark.timeout(120)
def test_rendezvous_with_eight_ranks_and_stray_duplicate_and_late_clients():
    """VERDICT r04 item 8 / ADVICE r04: rank 0 serves the id to seven peers while a port scanner connects and says nothing, a
    foreign client announces rank 99, one rank's first connection dies before the acknowledgement (it retries: served again, counted
    once), and one rank shows up two seconds late -- every rank ends with the same id and NativeComm joins 8 of 8."""
    import struct
    import threading
    import time
    from pyphysim_amd import distributed as D
    port = _free_port()
    world = 8
    ids, errors = {}, []
    _StubEngine.joined = []

    def rank_main(rank, delay=0.0, drop_first=False):
        try:
            time.sleep(delay)
            if drop_first:           # first attempt: send the rank, read the id, close WITHOUT acknowledging
                for _ in range(100):
                    try:
                        with socket.create_connection(("127.0.0.1", port), timeout=2.0) as c:
                            c.sendall(struct.pack("<i", rank))
                            c.recv(128)
                        break
                    except OSError:
                        time.sleep(0.05)
            eng = _StubEngine()
            comm = D.NativeComm(eng, rank=rank, world=world, master_addr="127.0.0.1", master_port=port, timeout=30.0)
            ids[rank] = eng.inits
            assert comm.rank == rank and comm.world == world
        except Exception as exc:     # noqa: BLE001 -- reported by the main thread
            errors.append((rank, repr(exc)))

    def stray(payload):
        for _ in range(100):
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=2.0) as c:
                    if payload:
                        c.sendall(payload)
                        try:
                            assert c.recv(128) == b""          # not one of ours: no id for it
                        except OSError:
                            pass
                return
            except OSError:
                time.sleep(0.05)

    threads = [threading.Thread(target=rank_main, args=(0,))]
    threads += [threading.Thread(target=rank_main, args=(r,)) for r in (1, 2, 3, 4)]
    threads += [threading.Thread(target=rank_main, args=(5,), kwargs=dict(drop_first=True))]
    threads += [threading.Thread(target=rank_main, args=(6,), kwargs=dict(delay=2.0))]
    threads += [threading.Thread(target=rank_main, args=(7,))]
    threads += [threading.Thread(target=stray, args=(b"",)), threading.Thread(target=stray, args=(struct.pack("<i", 99),)),
                threading.Thread(target=stray, args=(b"\x01",))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60.0)
    assert not errors, errors
    assert sorted(ids) == list(range(8)) and all(v == [(r, 8)] for r, v in ids.items())
    assert sorted(_StubEngine.joined) == list(range(8))
