"""Rank plumbing for realization-sharded runs: who am I, how many are we, and ONE all-reduce of the counters.

Two interchangeable back ends with the same small surface (rank, world, allreduce_counters, allreduce_floats,
broadcast_ints), used by simulations.BatchedSimulationRunner:

  NativeComm   libmcle's own RCCL communicator (csrc/comm.hip: mcle_comm_init / mcle_counters_allreduce) -- the
               route a C caller has.  The 128-byte RCCL id travels from rank 0 to the others over a plain TCP
               socket at MASTER_ADDR:MASTER_PORT (127.0.0.1 by default); torch.distributed is not involved.
  TorchComm    an initialised torch.distributed process group (backend nccl = RCCL on GPUs, gloo on CPU hosts
               and in the CPU tests).

The reference has nothing of the kind: its parallel mode farms whole parameter variations out to ipyparallel
engines (simulations/runner.py:1836-1846).
"""
import os
import socket
import struct
import time

import numpy as np

COUNTER_KEYS = ("n_realizations", "n_skipped", "sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq")


def env_rank_world():
    """(rank, local_rank, world) from the launcher's environment (torch.distributed.run, mpirun, srun)."""
    for r, l, w in (("RANK", "LOCAL_RANK", "WORLD_SIZE"), ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_LOCAL_RANK",
                                                          "OMPI_COMM_WORLD_SIZE"),
                    ("SLURM_PROCID", "SLURM_LOCALID", "SLURM_NTASKS")):
        if r in os.environ and w in os.environ:
            return int(os.environ[r]), int(os.environ.get(l, os.environ[r])), int(os.environ[w])
    return 0, 0, 1


def _exchange_id(make_id, rank, world, addr, port, timeout=120.0):
    """Rank 0 serves the id to the world - 1 peers that connect; returns the id on every rank.

    Protocol per connection: peer -> its rank (4 bytes); root -> the 128-byte id; peer -> ACK (0x06); root -> COMMIT (0x04).  The
    root counts a peer when it reads the ACK and closes its listener once every peer is counted; connections are served
    CONCURRENTLY (one slow or half-open client does not hold up the others).  A peer that has the id but never saw
    the COMMIT retries (the reply is idempotent); if the retry finds the listener GONE, the root has counted every peer --
    this one included -- and the peer returns the id it holds instead of failing while the root proceeds."""
    if world == 1:
        return make_id()
    if rank == 0:
        import threading
        payload = make_id()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(max(world, 16))
        seen, lock, all_in = set(), threading.Lock(), threading.Event()

        def serve(conn):
            with conn:
                try:                                  # one bad client (port scanner, half-open socket) must not
                    conn.settimeout(10.0)             # block or abort the others
                    head = b""
                    while len(head) < 4:
                        chunk = conn.recv(4 - len(head))
                        if not chunk:
                            return
                        head += chunk
                    peer = struct.unpack("<i", head)[0]
                    if not (1 <= peer < world):
                        return                        # not one of ours: no id for it
                    conn.sendall(payload)             # a rank that retries (its first reply was lost) is served again
                    if conn.recv(1) != b"\x06":       # counted when it ACKNOWLEDGES the id: sendall returning says nothing
                        return                        # about delivery
                    with lock:
                        seen.add(peer)
                        if len(seen) >= world - 1:
                            all_in.set()
                    conn.sendall(b"\x04")             # COMMIT: the peer may go on
                except (OSError, ConnectionError, struct.error):
                    return
        deadline = time.time() + timeout
        workers = []
        try:
            while not all_in.is_set():
                left = deadline - time.time()
                if left <= 0:
                    raise socket.timeout("rendezvous: %d of %d peers after %.0f s" % (len(seen), world - 1, timeout))
                srv.settimeout(min(left, 0.25))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                th = threading.Thread(target=serve, args=(conn,), daemon=True)
                th.start()
                workers.append(th)
        finally:
            srv.close()
        for th in workers:                            # let the COMMIT bytes of the last handlers go out
            th.join(timeout=2.0)
        return payload
    deadline = time.time() + timeout
    held = None                                       # the id of an attempt whose COMMIT never arrived
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.settimeout(10.0)
                conn.sendall(struct.pack("<i", rank))
                buf = b""
                while len(buf) < 128:
                    chunk = conn.recv(128 - len(buf))
                    if not chunk:
                        raise ConnectionError("short id")
                    buf += chunk
                conn.sendall(b"\x06")                      # acknowledge: only now does the root count this rank
                held = buf
                if conn.recv(1) == b"\x04":
                    return buf
                raise ConnectionError("no commit")
        except ConnectionRefusedError:
            if held is not None:                           # listener gone AFTER our acknowledgement went out: every peer is counted
                return held
            if time.time() > deadline:
                raise
            time.sleep(0.2)
        except (ConnectionError, OSError):
            if time.time() > deadline:
                if held is not None:
                    return held
                raise
            time.sleep(0.2)


class NativeComm:
    """libmcle's RCCL communicator on `engine`'s context."""

    def __init__(self, engine, rank=None, world=None, master_addr=None, master_port=None, unique_id=None, timeout=120.0):
        """unique_id: the 128-byte RCCL id when the caller has already distributed it (NativeComm.rendezvous, or any other
        channel); None = rendezvous here.  Two steps for callers that want to agree on the outcome of the first before
        entering the second: ncclCommInitRank blocks until EVERY rank has joined."""
        env_rank, _, env_world = env_rank_world()
        self.engine = engine
        self.rank = env_rank if rank is None else int(rank)
        self.world = env_world if world is None else int(world)
        uid = unique_id if unique_id is not None else self.rendezvous(engine, self.rank, self.world, master_addr, master_port,
                                                                       timeout)
        engine.comm_init(uid, self.rank, self.world)
        self._cnt = engine.new_counters()

    @staticmethod
    def rendezvous(engine, rank, world, master_addr=None, master_port=None, timeout=120.0):
        """Rank 0 draws the RCCL id and serves it over TCP at MASTER_ADDR : MCLE_COMM_PORT (default MASTER_PORT + 17);
        returns the id on every rank, raises after `timeout` seconds."""
        addr = master_addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(master_port or os.environ.get("MCLE_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 17))
        return _exchange_id(engine.comm_unique_id, int(rank), int(world), addr, port, timeout)

    def close(self):
        self.engine.comm_destroy()

    def allreduce_counters(self, c):
        """dict with COUNTER_KEYS + n_symbols / n_bits -> the same dict reduced over the ranks (exact integers)."""
        from . import _lib
        words = np.array([int(c.get(k, 0)) for k in _lib.COUNTER_FIELDS], dtype=np.uint64)
        self._cnt.set(words.view(self._cnt.dtype))
        self.engine.counters_allreduce(self._cnt, 1)
        out = self.engine.read_counters(self._cnt)
        return {k: int(out[k]) for k in _lib.COUNTER_FIELDS}

    def allreduce_floats(self, values):
        return [float(v) for v in self.engine.allreduce_f64(np.asarray(values, dtype=np.float64))]

    def broadcast_ints(self, values, src=0):
        """Rank `src`'s integer list on every rank: the uint64 SUM all-reduce of the counter blocks over a vector that is
        zero on the other ranks -- exact for every int64 (two's complement wraps back), unlike a float64 reduction, which
        rounds the squared-error sums of long runs above 2^53."""
        vals = [int(v) for v in values]
        n_blocks = max(1, -(-len(vals) // 6))
        words = np.zeros((n_blocks, 8), dtype=np.uint64)          # words 0..5 of a block are summed, 6..7 take the maximum
        if self.rank == src:
            flat = np.zeros(n_blocks * 6, dtype=np.uint64)
            flat[:len(vals)] = np.array([v & 0xFFFFFFFFFFFFFFFF for v in vals], dtype=np.uint64)
            words[:, :6] = flat.reshape(n_blocks, 6)
        cnt = self.engine.zeros(n_blocks, self._cnt.dtype)
        cnt.set(np.ascontiguousarray(words).view(self._cnt.dtype).reshape(n_blocks))
        self.engine.counters_allreduce(cnt, n_blocks)
        back = np.frombuffer(cnt.get().tobytes(), dtype=np.uint64).reshape(n_blocks, 8)[:, :6].reshape(-1)[:len(vals)]
        return [int(x) - (1 << 64) if int(x) >= (1 << 63) else int(x) for x in back]


class TorchComm:
    """An initialised torch.distributed process group (None = the default group)."""

    def __init__(self, process_group=None):
        import torch.distributed as dist
        self._dist, self.group = dist, process_group
        self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)

    def _device(self):
        import torch
        if self._dist.get_backend(self.group) == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def allreduce_counters(self, c):
        import torch
        dist = self._dist
        vec = torch.tensor([int(c[k]) for k in COUNTER_KEYS], dtype=torch.int64, device=self._device())
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group)              # the path's only exchange
        shape = torch.tensor([int(c.get("n_symbols", 0)), int(c.get("n_bits", 0))], dtype=torch.int64, device=self._device())
        dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=self.group)            # ranks with an empty shard
        out = {k: int(v) for k, v in zip(COUNTER_KEYS, vec.tolist())}
        out["n_symbols"], out["n_bits"] = int(shape[0]), int(shape[1])
        return out

    def allreduce_floats(self, values):
        import torch
        vec = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self._device())
        self._dist.all_reduce(vec, op=self._dist.ReduceOp.SUM, group=self.group)
        return [float(v) for v in vec.tolist()]

    def broadcast_ints(self, values, src=0):
        import torch
        vec = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self._device())
        self._dist.broadcast(vec, src=src, group=self.group)
        return [int(v) for v in vec.tolist()]


def default_comm(process_group=None):
    """TorchComm when torch.distributed is initialised, else None (single rank)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    if dist.is_available() and dist.is_initialized():
        return TorchComm(process_group)
    return None
