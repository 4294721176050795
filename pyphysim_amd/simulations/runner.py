"""SimulationRunner: the Monte Carlo repetition loop of the reference (simulations/runner.py:1076-1948)
with the same overridables, plus BatchedSimulationRunner, which advances the loop by whole GPU
batches of realizations and (optionally) shards every batch over torch.distributed ranks.

Kept surface: rep_max, params, results, runned_reps, elapsed_time, set_results_filename,
partial-result save / resume, progressbar_message, update_progress_function_style,
_run_simulation, _keep_going, _on_simulate_start / finish, _on_simulate_current_params_start /
finish, simulate(param_variation_index=None), SkipThisOne.  Not carried over (out of the hot
path): configobj files, command-line parsing, ipyparallel,
progress-bar widgets.
"""
import os
import pickle
import time

from .parameters import SimulationParameters
from .results import Result, SimulationResults


def get_partial_results_filename(results_base_filename, current_params, partial_results_folder=None):
    """'<base>_unpack_<index, zero-padded to the digits of the number of variations>.pickle' inside
    `partial_results_folder` (reference simulations/runner.py:109-145)."""
    total_unpacks = current_params.get_num_unpacked_variations()
    num_digits = len(str(total_unpacks))
    unpack_index_str = str(current_params.unpack_index).zfill(num_digits)
    name = "{0}_unpack_{1}.pickle".format(results_base_filename, unpack_index_str)
    if partial_results_folder is not None:
        name = os.path.join(partial_results_folder, name)
    return name


class SkipThisOne(Exception):
    """Raise inside _run_simulation to discard the current realization (runner.py:151-185)."""

    def __init__(self, msg):
        super().__init__()
        self.msg = msg

    def __str__(self):
        return "SkipThisOne: {0}".format(self.msg)


def _pretty_time(seconds):
    seconds = int(round(seconds))
    h, rem = divmod(seconds, 3600)
    m, s = divmod(rem, 60)
    if h:
        return "%dh:%02dm:%02ds" % (h, m, s)
    if m:
        return "%dm:%02ds" % (m, s)
    return "%ds" % s


class SimulationRunner:
    def __init__(self, default_config_file=None, config_spec=None, read_command_line_args=True,
                 save_parsed_file=False):
        if default_config_file is not None:
            raise NotImplementedError("configobj parameter files are outside this package's scope; fill "
                                      "runner.params programmatically")
        self.rep_max = 1
        self._runned_reps = []
        self._params = SimulationParameters()
        self._results = SimulationResults()
        self._results_filename = None
        self.delete_partial_results_bool = False
        self.partial_results_folder = "partial_results"
        self.progressbar_message = "Progress"
        self.update_progress_function_style = "text2"
        self.progress_output_type = "screen"
        self.partial_save_every_reps = 500        # runner.py:109-145
        self.partial_save_every_seconds = 300.0
        self._last_partial_save = 0.0
        self._tic = self._toc = 0.0
        self._partial_files = []

    # ---- properties of the reference --------------------------------------------------------
    params = property(lambda self: self._params)
    results = property(lambda self: self._results)
    runned_reps = property(lambda self: self._runned_reps)
    results_filename = property(lambda self: None if self._results_filename is None
                                else self._results_filename + ".pickle")

    @property
    def elapsed_time(self):
        return _pretty_time(self._toc - self._tic)

    def __repr__(self):
        return "{0}(rep_max={1}, num_params_variations={2})".format(
            self.__class__.__name__, self.rep_max, self.params.get_num_unpacked_variations())

    def set_results_filename(self, filename=None):
        """Base name (no extension); '{param}' fields are replaced when saving (runner.py:1216)."""
        self._results_filename = filename

    def clear(self):
        self._runned_reps = []
        self._results = SimulationResults()
        self._partial_files = []

    # ---- what a simulator overrides ---------------------------------------------------------
    def _run_simulation(self, current_parameters):
        raise NotImplementedError("'_run_simulation' must be implemented in a subclass of SimulationRunner")

    def _keep_going(self, current_params, current_sim_results, current_rep):
        return True

    def _on_simulate_start(self):
        pass

    def _on_simulate_finish(self):
        pass

    def _on_simulate_current_params_start(self, current_params):
        pass

    def _on_simulate_current_params_finish(self, current_params, current_params_sim_results):
        pass

    # ---- partial results (runner.py:926-1069) -----------------------------------------------
    def _partial_name(self, current_params):
        if self._results_filename is None:
            return None
        base = self._results.get_filename_with_replaced_params(os.path.basename(self._results_filename))
        folder = os.path.join(os.path.dirname(self._results_filename) or ".", self.partial_results_folder)
        return get_partial_results_filename(base, current_params, folder)

    def _save_partial(self, current_rep, current_params, current_sim_results):
        name = self._partial_name(current_params)
        if name is None:
            return None
        os.makedirs(os.path.dirname(name), exist_ok=True)
        current_sim_results.current_rep = current_rep
        current_sim_results.set_parameters(current_params)
        with open(name, "wb") as fh:
            pickle.dump(current_sim_results, fh, protocol=2)
        self._last_partial_save = time.time()
        if name not in self._partial_files:
            self._partial_files.append(name)
        return name

    def _save_partial_maybe(self, current_rep, current_params, current_sim_results):
        if self._results_filename is None:
            return
        if (current_rep % self.partial_save_every_reps == 0
                or time.time() - self._last_partial_save > self.partial_save_every_seconds):
            self._save_partial(current_rep, current_params, current_sim_results)

    def _load_partial(self, current_params):
        name = self._partial_name(current_params)
        if name is None or not os.path.exists(name):
            return None
        with open(name, "rb") as fh:
            partial = pickle.load(fh)
        if partial.params != current_params:
            raise ValueError("Partial results loaded from file does not match current parameters. \n"
                             "File: {0}".format(name))
        return partial

    # ---- the repetition loop (runner.py:1435-1539) ------------------------------------------
    def _timed_run(self, current_params):
        tic = time.time()
        res = self._run_simulation(current_params)
        res.add_result(Result.create("elapsed_time", Result.SUMTYPE, time.time() - tic))
        return res

    def _simulate_for_current_params(self, current_params):
        self._on_simulate_current_params_start(current_params)
        current_sim_results = self._load_partial(current_params)
        if current_sim_results is None:
            # NB: like the reference, the first repetition is outside the SkipThisOne guard
            current_sim_results = self._timed_run(current_params)
            current_rep = 1
        else:
            current_rep = current_sim_results.current_rep
        current_sim_results.add_new_result("num_skipped_reps", Result.SUMTYPE, 0)
        while self._keep_going(current_params, current_sim_results, current_rep) and current_rep < self.rep_max:
            try:
                current_sim_results.merge_all_results(self._timed_run(current_params))
                current_rep += 1
            except SkipThisOne:
                current_sim_results["num_skipped_reps"][-1].update(1)
            self._save_partial_maybe(current_rep, current_params, current_sim_results)
        self._on_simulate_current_params_finish(current_params, current_sim_results)
        self._save_partial(current_rep, current_params, current_sim_results)
        return current_rep, current_sim_results

    def _common_setup(self):
        self.clear()
        self.params.parameters.setdefault("rep_max", self.rep_max)
        self._results.set_parameters(self.params)
        self._tic = time.time()
        self._last_partial_save = time.time()
        self._on_simulate_start()

    def _common_cleanup(self):
        self._on_simulate_finish()
        self._toc = time.time()
        self._results.runned_reps = self._runned_reps
        if self._results_filename is not None:
            self._results.save_to_file(self._results_filename + ".pickle")
            if self.delete_partial_results_bool:
                for name in self._partial_files:
                    if os.path.exists(name):
                        os.remove(name)
                folder = os.path.join(os.path.dirname(self._results_filename) or ".", self.partial_results_folder)
                if os.path.isdir(folder) and not os.listdir(folder):
                    os.rmdir(folder)

    def simulate(self, param_variation_index=None):
        """All parameter variations (or only variation `param_variation_index`, whose partial results
        are stored for a later combine step; that mode needs a results filename)."""
        self._common_setup()
        variations = self.params.get_unpacked_params_list()
        if param_variation_index is None:
            for current_params in variations:
                reps, res = self._simulate_for_current_params(current_params)
                self._runned_reps.append(reps)
                self._results.append_all_results(res)
            self._common_cleanup()
            return
        if self._results_filename is None:
            raise RuntimeError('The results filename must be set before calling the "simulate" method.')
        index = int(param_variation_index)
        if 0 <= index < len(variations):
            reps, _ = self._simulate_for_current_params(variations[index])
            self._runned_reps = reps
        self._toc = time.time()


    # ---- the reference's task-parallel mode (runner.py:1774-1886) -----------------------------------------------
    def simulate_common_cleaning(self):
        """runner.py:1621-1634."""
        self._common_cleanup()

    @staticmethod
    def _simulate_for_current_params_parallel(obj, current_params, update_progress_func=None):
        """runner.py:1541-1619: what one engine of the parallel view runs -> (reps, results, partial file name)."""
        reps, res = obj._simulate_for_current_params(current_params)
        name = obj._partial_name(current_params) if obj._results_filename is not None else None
        return reps, res, name

    def simulate_in_parallel(self, view=None, wait=True):
        """runner.py:1774-1855: one parameter variation per engine of an ipyparallel-style `view` (anything with
        `map(func, *iterables, block=False)` whose return value has `wait()` and `get()`).  Without a view the
        reference starts a local ipyparallel cluster; here the variations then simply run one after the other through
        `simulate()` -- on this engine the parallel axis is the realization index (a BatchedSimulationRunner shards
        every variation over the ranks of its process group), not the parameter variation."""
        if view is None:
            self.simulate()
            self._async_results = None
            return
        self._common_setup()
        variations = self.params.get_unpacked_params_list()
        self._async_results = view.map(SimulationRunner._simulate_for_current_params_parallel, [self] * len(variations),
                                       variations, [None] * len(variations), block=False)
        if wait:
            self.wait_parallel_simulation()

    def wait_parallel_simulation(self):
        """runner.py:1857-1886."""
        pending = getattr(self, "_async_results", None)
        if pending is None:
            return
        pending.wait()
        for reps, res, name in pending.get():
            self._runned_reps.append(reps)
            self._results.append_all_results(res)
            if name is not None and name not in self._partial_files:
                self._partial_files.append(name)
        self.simulate_common_cleaning()
        self._async_results = None


class BatchedSimulationRunner(SimulationRunner):
    """Repetition loop in units of GPU batches.

    A subclass implements ``_run_batch(current_parameters, first_rep, count) -> dict`` returning the
    integer counter block of realizations [first_rep, first_rep + count) (the dict produced by
    ``Engine.run_*``: n_realizations, n_skipped, sym_errors, sym_errors_sq, bit_errors,
    bit_errors_sq, n_symbols, n_bits; plus any ``EXTRA_KEYS`` it declares, summed like the counters)
    and may override ``_results_from_counters``.

    * realization index == repetition index: results depend only on (seed, index), never on the
      batch size, the number of ranks or a resume point;
    * with more than one rank (``comm=``: a pyphysim_amd.distributed.NativeComm / TorchComm; default: the
      initialised torch.distributed group, if any) every batch is split contiguously over the ranks and the
      counters are all-reduced (exact integers, order independent) ONCE per parameter variation -- every rank
      runs its share of all the batches first.  Only a simulator that overrides ``_keep_going`` needs the
      global counters to decide whether to go on, and reduces after every batch;
    * skipped realizations (singular channel etc.) count into 'num_skipped_reps' and are replaced
      by later indices, like SkipThisOne in the serial loop (one more round of batches, and one more
      reduction, per round of replacements);
    * partial results are written and removed by rank 0 only; a resume point read by rank 0 is broadcast.
    """
    COUNTER_KEYS = ("n_realizations", "n_skipped", "sym_errors", "sym_errors_sq", "bit_errors", "bit_errors_sq")
    EXTRA_KEYS = ()          # further per-batch sums a subclass carries (floats or integers), reduced as float64
    EXTRA_INT_KEYS = ()      # the subset of EXTRA_KEYS that holds integers

    def __init__(self, batch_size=4096, process_group=None, exact_early_stop=False, comm=None):
        super().__init__(read_command_line_args=False)
        self.batch_size = int(batch_size)
        self.process_group = process_group
        self.comm = comm
        self.first_rep = 0
        # True: a custom _keep_going is consulted after EVERY realization, like the reference's loop
        # (runner.py:1491), by replaying the batch's per-realization counts on the host; the batch is
        # cut at the realization where the rule first says stop (single rank only).
        self.exact_early_stop = bool(exact_early_stop)
        # a configuration whose every realization is skipped (e.g. zero forcing on a rank-deficient set-up)
        # would never reach rep_max: give up after this many consecutive batches without one good realization
        self.max_all_skipped_batches = 64
        self.n_reductions = 0    # all-reduces issued by the last simulate() (diagnostics, tests)

    def _run_batch(self, current_parameters, first_rep, count):
        raise NotImplementedError("'_run_batch' must be implemented in a subclass of BatchedSimulationRunner")

    def _run_batch_detailed(self, current_parameters, first_rep, count):
        """-> (counters, sym_err[count], bit_err[count]) with 0xFFFFFFFF marking skipped realizations;
        needed only for exact_early_stop."""
        raise NotImplementedError("exact_early_stop needs '_run_batch_detailed'")

    def _replay(self, current_params, total, elapsed, c, se, be):
        """Feed the batch realization by realization; returns (totals, stopped)."""
        n_sym, n_bits = c["n_symbols"], c["n_bits"]
        for s, b in zip(se.tolist(), be.tolist()):
            one = self._zero_like(c)
            if s == 0xFFFFFFFF:
                one["n_skipped"] = 1
            else:
                one.update(n_realizations=1, sym_errors=s, sym_errors_sq=s * s, bit_errors=b, bit_errors_sq=b * b)
            one["n_symbols"], one["n_bits"] = n_sym, n_bits
            total = self._add(total, one)
            if total["n_realizations"] >= self.rep_max:
                return total, True
            if s != 0xFFFFFFFF and not self._keep_going(current_params,
                                                        self._finish_results(current_params, total, elapsed),
                                                        total["n_realizations"]):
                return total, True
        return total, False

    def _run_simulation(self, current_parameters):
        c = self._run_batch(current_parameters, self.first_rep, 1)
        return self._results_from_counters(current_parameters, c)

    def _results_from_counters(self, current_parameters, c):
        """The Result set of the reference's link simulators (apps/awgn_modulators/simulate_psk.py:
        90-112, apps/mimo/simulate_mimo.py:117-139)."""
        n = c["n_realizations"]
        res = SimulationResults()
        res.add_result(Result.from_counters("symbol_errors", Result.SUMTYPE, c["sym_errors"], c["sym_errors_sq"], 1, n))
        res.add_result(Result.from_batch("num_symbols", Result.SUMTYPE, c["n_symbols"] * n, 0, float(c["n_symbols"] * n),
                                         float(c["n_symbols"]) ** 2 * n, n))
        res.add_result(Result.from_counters("bit_errors", Result.SUMTYPE, c["bit_errors"], c["bit_errors_sq"], 1, n))
        res.add_result(Result.from_batch("num_bits", Result.SUMTYPE, c["n_bits"] * n, 0, float(c["n_bits"] * n),
                                         float(c["n_bits"]) ** 2 * n, n))
        res.add_result(Result.from_counters("ber", Result.RATIOTYPE, c["bit_errors"], c["bit_errors_sq"], c["n_bits"], n))
        res.add_result(Result.from_counters("ser", Result.RATIOTYPE, c["sym_errors"], c["sym_errors_sq"],
                                            c["n_symbols"], n))
        return res

    # ---- sharding ---------------------------------------------------------------------------
    def _comm(self):
        """-> (comm or None, rank, world)"""
        comm = self.comm
        if comm is None:
            from ..distributed import default_comm
            comm = default_comm(self.process_group)
        if comm is None:
            return None, 0, 1
        return comm, comm.rank, comm.world

    @staticmethod
    def shard_range(first, count, rank, world):
        """Contiguous split of [first, first + count) -> (first_r, count_r) of `rank`."""
        lo = first + (count * rank) // world
        hi = first + (count * (rank + 1)) // world
        return lo, hi - lo

    def _allreduce(self, c):
        comm, rank, world = self._comm()
        if world == 1:
            return c
        self.n_reductions += 1
        out = comm.allreduce_counters(c)                 # the path's only exchange: exact integer sums
        if self.EXTRA_KEYS:
            vals = comm.allreduce_floats([float(c.get(k, 0)) for k in self.EXTRA_KEYS])
            for k, v in zip(self.EXTRA_KEYS, vals):
                out[k] = int(round(v)) if k in self.EXTRA_INT_KEYS else v
        return out

    def _add(self, acc, c):
        if acc is None:
            return dict(c)
        for k in self.COUNTER_KEYS + tuple(self.EXTRA_KEYS):
            acc[k] = acc.get(k, 0) + c.get(k, 0)
        acc["n_symbols"] = max(acc.get("n_symbols", 0), c.get("n_symbols", 0))
        acc["n_bits"] = max(acc.get("n_bits", 0), c.get("n_bits", 0))
        return acc

    def _zero_like(self, c):
        z = {k: 0 for k in self.COUNTER_KEYS + tuple(self.EXTRA_KEYS)}
        z["n_symbols"] = 0 if c is None else c["n_symbols"]
        z["n_bits"] = 0 if c is None else c["n_bits"]
        return z

    # ---- files: rank 0 only -----------------------------------------------------------------
    def _common_setup(self):
        self.n_reductions = 0
        super()._common_setup()

    def _common_cleanup(self):
        _, rank, _ = self._comm()
        if rank == 0:
            return super()._common_cleanup()
        self._on_simulate_finish()
        self._toc = time.time()
        self._results.runned_reps = self._runned_reps

    def _resume_state(self, current_params):
        """(total, next_index, elapsed) of a partial-results file -- read by rank 0, the same on every rank."""
        comm, rank, world = self._comm()
        state, error = None, None
        if rank == 0:
            try:
                partial = self._load_partial(current_params)
            except ValueError as exc:                    # parameters of the file do not match (runner.py:1058-1063)
                if world == 1:
                    raise
                partial, error = None, exc
            if partial is not None:
                state = partial._batched_state
        if world == 1:
            return state
        keys = self.COUNTER_KEYS + ("n_symbols", "n_bits")
        if rank == 0 and error is not None:
            ints = [2, 0] + [0] * len(keys)
            floats = [0.0] * (1 + len(self.EXTRA_KEYS))
        elif rank == 0 and state is not None:
            ints = [1, int(state[1])] + [int(state[0].get(k, 0)) for k in keys]
            floats = [float(state[2])] + [float(state[0].get(k, 0)) for k in self.EXTRA_KEYS]
        else:
            ints = [0, 0] + [0] * len(keys)
            floats = [0.0] * (1 + len(self.EXTRA_KEYS))
        ints = comm.broadcast_ints(ints, src=0)
        floats = comm.allreduce_floats(floats)           # zero everywhere but rank 0
        if ints[0] == 2:                                 # every rank fails together instead of deadlocking
            raise error if error is not None else ValueError(
                "Partial results loaded from file does not match current parameters (reported by rank 0)")
        if not ints[0]:
            return None
        total = dict(zip(keys, ints[2:]))
        for k, v in zip(self.EXTRA_KEYS, floats[1:]):
            total[k] = int(round(v)) if k in self.EXTRA_INT_KEYS else v
        return total, ints[1], floats[0]

    def _save_state(self, current_params, total, next_index, elapsed):
        snap = self._finish_results(current_params, total, elapsed)
        snap._batched_state = (dict(total), next_index, elapsed)
        self._save_partial(total["n_realizations"], current_params, snap)

    # ---- the batched loop -------------------------------------------------------------------
    def _simulate_for_current_params(self, current_params):
        self._on_simulate_current_params_start(current_params)
        comm, rank, world = self._comm()
        state = self._resume_state(current_params)
        if state is not None:
            total, next_index, elapsed = dict(state[0]), state[1], state[2]
        else:
            total, next_index, elapsed = self._zero_like(None), self.first_rep, 0.0
        custom_stop = type(self)._keep_going is not SimulationRunner._keep_going
        exact = self.exact_early_stop and custom_stop
        if exact and world > 1:
            raise RuntimeError("exact_early_stop replays realizations in index order and is single-rank only")
        saving = self._results_filename is not None and rank == 0
        last_saved_reps = total["n_realizations"]
        barren = 0           # consecutive batches without a single good realization
        stopped = False
        while total["n_realizations"] < self.rep_max and not stopped:
            if exact:
                want = min(self.batch_size, self.rep_max - total["n_realizations"])
                tic = time.time()
                c, se, be = self._run_batch_detailed(current_params, next_index, want)
                elapsed += time.time() - tic
                next_index += want
                before = total["n_realizations"]
                total, stopped = self._replay(current_params, total, elapsed, c, se, be)
                barren = 0 if total["n_realizations"] > before else barren + 1
            else:
                if custom_stop and total["n_realizations"] > 0:
                    # the reference evaluates _keep_going before every repetition after the first
                    # (runner.py:1491); here it is evaluated before every batch after the first
                    snapshot = self._finish_results(current_params, total, elapsed)
                    if not self._keep_going(current_params, snapshot, total["n_realizations"]):
                        break
                # one round: every batch still owed (a single one when a stopping rule needs global numbers)
                owed = self.rep_max - total["n_realizations"]
                n_batches = 1 if custom_stop else -(-owed // (self.batch_size * world))
                if world > 1 and self._results_filename is not None:
                    # a multi-rank run can only write a resume point after a reduction (the totals are global then), so with
                    # a results file a round ends after about partial_save_every_reps realizations instead of running every
                    # owed batch: a crash loses one round, not the variation.  Decided from the realization count alone --
                    # every rank takes the same decision (a wall-clock rule would not).
                    n_batches = min(n_batches, max(1, -(-self.partial_save_every_reps // (self.batch_size * world))))
                local = self._zero_like(None)
                for _ in range(n_batches):
                    want = min(self.batch_size * world, owed)
                    lo, cnt = self.shard_range(next_index, want, rank, world)
                    tic = time.time()
                    if cnt > 0:
                        local = self._add(local, self._run_batch(current_params, lo, cnt))
                    elapsed += time.time() - tic
                    next_index += want
                    owed -= want
                    if saving and world == 1 and (
                            total["n_realizations"] + local["n_realizations"] - last_saved_reps >= self.partial_save_every_reps
                            or time.time() - self._last_partial_save > self.partial_save_every_seconds):
                        # single rank: the running totals are global, a resume point can be written mid-round
                        so_far = self._add(dict(total), local)
                        self._save_state(current_params, so_far, next_index, elapsed)
                        last_saved_reps = so_far["n_realizations"]
                got = self._allreduce(local)
                barren = 0 if got["n_realizations"] > 0 else barren + n_batches
                total = self._add(total, got)
                if saving and (total["n_realizations"] - last_saved_reps >= self.partial_save_every_reps
                               or time.time() - self._last_partial_save > self.partial_save_every_seconds):
                    self._save_state(current_params, total, next_index, elapsed)
                    last_saved_reps = total["n_realizations"]
            if barren >= self.max_all_skipped_batches:
                raise RuntimeError("every realization of the last %d batches was skipped (%d skipped in total): this "
                                   "configuration cannot produce a valid realization" % (barren, total["n_skipped"]))
        final = self._finish_results(current_params, total, elapsed)
        self._on_simulate_current_params_finish(current_params, final)
        if rank == 0:
            final._batched_state = (dict(total), next_index, elapsed)
            self._save_partial(total["n_realizations"], current_params, final)
        return total["n_realizations"], final

    def _finish_results(self, current_params, total, elapsed):
        n = max(int(total["n_realizations"]), 1)
        res = self._results_from_counters(current_params, total)
        # one 'elapsed_time' update per realization in the reference; here the batch time is spread evenly
        res.add_result(Result.from_batch("elapsed_time", Result.SUMTYPE, elapsed, 0, elapsed, elapsed * elapsed / n, n))
        res.add_result(Result.from_batch("num_skipped_reps", Result.SUMTYPE, int(total["n_skipped"]), 0,
                                         float(total["n_skipped"]), float(total["n_skipped"]),
                                         1 + int(total["n_skipped"])))
        return res
