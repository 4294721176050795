"""Result / SimulationResults: the accumulators every realization's outcome lands in
(reference simulations/results.py:128-786 and :795-1627).

Same state as the reference -- value, total, running sum and sum of squares of the per-update
result, number of updates -- so means, variances and confidence intervals come out identical.
New here: :meth:`Result.from_batch` / :meth:`Result.from_counters` materialise the state that
`count` sequential ``update`` calls would have produced from the integer counter block a GPU batch
returns (the recipe of the reference's own ``Result._from_dict``, results.py:747-786).
"""
import json
import os
import pickle
from collections.abc import Iterable

import numpy as np

from .parameters import (SimulationParameters, _json_default, _json_object_hook, combine_simulation_parameters,
                         replace_dict_values)

_CI_TABLE = {50: 0.674, 60: 0.842, 70: 1.036, 80: 1.282, 90: 1.645, 95: 1.96, 98: 2.326, 99: 2.576, 99.5: 2.807,
             99.8: 3.090, 99.9: 3.291}


def calc_confidence_interval(mean, std, n, P=95.0):
    """Normal-approximation interval (reference util/misc.py:807-867, same table of z values)."""
    norm_std = std / np.sqrt(n)
    return (mean - _CI_TABLE[P] * norm_std, mean + _CI_TABLE[P] * norm_std)


class Result:
    SUMTYPE, RATIOTYPE, MISCTYPE, CHOICETYPE = range(4)
    _all_types_names = {SUMTYPE: "SUMTYPE", RATIOTYPE: "RATIOTYPE", MISCTYPE: "MISCTYPE", CHOICETYPE: "CHOICETYPE"}

    def __init__(self, name, update_type_code, accumulate_values=False, choice_num=None):
        self.name = name
        self._update_type_code = update_type_code
        self._value = 0
        self._total = 0
        self._result_sum = 0.0
        self._result_squared_sum = 0.0
        self.num_updates = 0
        if update_type_code == Result.CHOICETYPE:
            if not isinstance(choice_num, (int, np.integer)):
                raise RuntimeError("'choice_num' argument for the Result object must be an integer for the "
                                   "CHOICETYPE type.")
            self._value = np.zeros(choice_num, dtype=int)
        self._accumulate_values_bool = accumulate_values
        self._value_list = []
        self._total_list = []

    # ---- construction -----------------------------------------------------------------------
    @staticmethod
    def create(name, update_type, value, total=0, accumulate_values=False):
        if update_type == Result.CHOICETYPE:
            if total == 0:
                raise RuntimeError("When creating a new Result of CHOICETYPE you must provide the 'total' as "
                                   "well as the 'value.")
            res = Result(name, update_type, accumulate_values, choice_num=total)
            res.update(value)
        else:
            res = Result(name, update_type, accumulate_values)
            res.update(value, total)
        return res

    @staticmethod
    def from_batch(name, update_type, value, total, result_sum, result_squared_sum, num_updates):
        """State after `num_updates` updates whose values sum to `value` (totals to `total`) and
        whose per-update results sum to `result_sum` (squares to `result_squared_sum`)."""
        if update_type not in (Result.SUMTYPE, Result.RATIOTYPE):
            raise ValueError("from_batch supports SUMTYPE and RATIOTYPE")
        res = Result(name, update_type)
        res._value, res._total = value, total
        res._result_sum, res._result_squared_sum = result_sum, result_squared_sum
        res.num_updates = int(num_updates)
        return res

    @staticmethod
    def from_counters(name, update_type, err_sum, err_sq_sum, units_per_update, num_updates):
        """From exact integer sums of per-realization error counts e_r: SUM -> value = sum e_r;
        RATIO(e_r, units) -> value/total plus sum and sum of squares of e_r/units."""
        if update_type == Result.SUMTYPE:
            return Result.from_batch(name, update_type, int(err_sum), 0, float(err_sum), float(err_sq_sum),
                                     num_updates)
        u = float(units_per_update)
        return Result.from_batch(name, update_type, int(err_sum), int(units_per_update) * int(num_updates),
                                 err_sum / u, err_sq_sum / (u * u), num_updates)

    # ---- accessors --------------------------------------------------------------------------
    accumulate_values_bool = property(lambda self: self._accumulate_values_bool)
    type_name = property(lambda self: Result._all_types_names[self._update_type_code])
    type_code = property(lambda self: self._update_type_code)

    def __repr__(self):
        if self._update_type_code == Result.RATIOTYPE:
            v, t = self._value, self._total
            if t != 0:
                return "Result -> {0}: {1}/{2} -> {3}".format(self.name, v, t, v / t)
            return "Result -> {0}: {1}/{2} -> NaN".format(self.name, v, t)
        return "Result -> {0}: {1}".format(self.name, self.get_result())

    def __eq__(self, other):
        if self is other:
            return True
        if not isinstance(other, Result):
            return False
        for att in ("name", "_update_type_code", "_total", "_accumulate_values_bool", "_value_list", "_total_list",
                    "_result_squared_sum", "_result_sum"):
            if getattr(self, att) != getattr(other, att):
                return False
        if self._update_type_code == Result.CHOICETYPE:
            return bool(np.array_equal(self._value, other._value))
        return bool(self._value == other._value)

    def __ne__(self, other):
        return not self.__eq__(other)

    # ---- update / merge (results.py:469-623) ------------------------------------------------
    def update(self, value, total=None):
        self.num_updates += 1
        code = self._update_type_code
        if code == Result.SUMTYPE:
            self._value += value
            self._result_sum += value
            self._result_squared_sum += value ** 2
            if self._accumulate_values_bool:
                self._value_list.append(value)
        elif code == Result.RATIOTYPE:
            if total is None:
                raise ValueError("A 'p_value' and a 'p_total' are required when updating a Result object of the "
                                 "RATIOTYPE type.")
            self._value += value
            self._total += total
            ratio = value / total
            self._result_sum += ratio
            self._result_squared_sum += ratio ** 2
            if self._accumulate_values_bool:
                self._value_list.append(value)
                self._total_list.append(total)
        elif code == Result.MISCTYPE:
            self._value = value
            if self._accumulate_values_bool:
                self._value_list.append(value)
        elif code == Result.CHOICETYPE:
            assert isinstance(value, (int, np.integer)), "Value for the CHOICETYPE must be an integer."
            self._value[value] += 1
            self._total += 1
            if self._accumulate_values_bool:
                self._value_list.append(value)
        else:
            raise ValueError("Can't update a Result object of type '{0}'".format(code))

    def merge(self, other):
        assert isinstance(other, Result)
        assert self._update_type_code == other._update_type_code, \
            "Can only merge two objects with the same name and type"
        assert self.name == other.name, "Can only merge two objects with the same name and type"
        if self._accumulate_values_bool:
            assert other._accumulate_values_bool, "The merged Result also must have been set to accumulate values."
            self._value_list.extend(other._value_list)
            self._total_list.extend(other._total_list)
        if self._update_type_code == Result.MISCTYPE:
            self.num_updates = other.num_updates
            self._value, self._total = other._value, other._total
            self._result_sum, self._result_squared_sum = other._result_sum, other._result_squared_sum
        else:
            self.num_updates += other.num_updates
            self._value = self._value + other._value
            self._total = self._total + other._total
            self._result_sum += other._result_sum
            self._result_squared_sum += other._result_squared_sum

    # ---- statistics -------------------------------------------------------------------------
    def get_result(self):
        if self.num_updates == 0:
            return "Nothing yet"
        if self._update_type_code in (Result.RATIOTYPE, Result.CHOICETYPE):
            return self._value / self._total
        return self._value

    def get_result_accumulated_values(self):
        return self._value_list

    def get_result_accumulated_totals(self):
        return self._total_list

    def get_result_mean(self):
        return self._result_sum / self.num_updates

    def get_result_var(self):
        return self._result_squared_sum / self.num_updates - self.get_result_mean() ** 2

    def get_confidence_interval(self, P=95.0):
        if self._update_type_code == Result.MISCTYPE:
            raise RuntimeError("Calling get_confidence_interval is not valid for the MISC update type.")
        return calc_confidence_interval(self.get_result_mean(), np.sqrt(self.get_result_var()), self.num_updates, P)

    # ---- (de)serialisation ------------------------------------------------------------------
    def to_dict(self):
        return dict(name=self.name, update_type_code=self._update_type_code, value=self._value, total=self._total,
                    result_sum=self._result_sum, result_squared_sum=self._result_squared_sum,
                    num_updates=self.num_updates, accumulate_values_bool=self._accumulate_values_bool,
                    value_list=self._value_list, total_list=self._total_list)

    _to_dict = to_dict

    @staticmethod
    def from_dict(d):
        if d["update_type_code"] == Result.CHOICETYPE and isinstance(d["value"], Iterable):
            values = list(d["value"])
            res = Result(d["name"], d["update_type_code"], d["accumulate_values_bool"], choice_num=len(values))
            res._value = np.asarray(values, dtype=int)
            res._total = d["total"]
        else:
            res = Result(d["name"], d["update_type_code"], d["accumulate_values_bool"])
            res._value, res._total = d["value"], d["total"]
        res._value_list, res._total_list = list(d["value_list"]), list(d["total_list"])
        res.num_updates = d["num_updates"]
        res._result_sum, res._result_squared_sum = d["result_sum"], d["result_squared_sum"]
        return res

    _from_dict = from_dict

    def to_json(self):
        return json.dumps(self.to_dict(), default=_json_default)

    @staticmethod
    def from_json(text):
        return Result.from_dict(json.loads(text, object_hook=_json_object_hook))


class SimulationResults:
    """name -> list of Result (one per parameter variation), plus the parameters they belong to."""

    def __init__(self):
        self._results = {}
        self._params = SimulationParameters()
        self.runned_reps = None
        self.original_filename = None
        self.current_rep = -1

    params = property(lambda self: self._params)

    def set_parameters(self, params):
        if not isinstance(params, SimulationParameters):
            raise ValueError("params must be a SimulationParameters object")
        self._params = params

    def __repr__(self):
        return "SimulationResults: {0}".format(sorted(self._results.keys()))

    def __eq__(self, other):
        if self is other:
            return True
        if not isinstance(other, SimulationResults):
            return False
        if self._params != other._params or self.runned_reps != other.runned_reps:
            return False
        if self._results.keys() != other._results.keys():
            return False
        return all(self[k] == other[k] for k in self._results if k != "elapsed_time")

    def __ne__(self, other):
        return not self.__eq__(other)

    # ---- building ---------------------------------------------------------------------------
    def add_result(self, result):
        self._results[result.name] = [result]

    def add_new_result(self, name, update_type, value, total=0):
        self.add_result(Result.create(name, update_type, value, total))

    def append_result(self, result):
        if result.name in self._results:
            if self._results[result.name][0].type_code != result.type_code:
                raise ValueError("Can only append to results of the same type")
            self._results[result.name].append(result)
        else:
            self.add_result(result)

    def append_all_results(self, other):
        for results in other:
            for result in results:
                self.append_result(result)

    def merge_all_results(self, other):
        """Merge the LAST Result of every name (results.py:1103-1159); 'num_skipped_reps' is
        created on demand."""
        if len(self) == 0:
            for name in other.get_result_names():
                self._results[name] = other[name]
            return
        for name in self.get_result_names():
            if name != "num_skipped_reps":
                self._results[name][-1].merge(other[name][-1])
        if "num_skipped_reps" in other.get_result_names():
            if "num_skipped_reps" not in self._results:
                self.add_new_result("num_skipped_reps", Result.SUMTYPE, 0)
            self._results["num_skipped_reps"][-1].merge(other["num_skipped_reps"][-1])

    # ---- reading ----------------------------------------------------------------------------
    def get_result_names(self):
        return list(self._results.keys())

    def __getitem__(self, key):
        return self._results[key]

    def __len__(self):
        return len(self._results)

    def __iter__(self):
        return iter(self._results.values())

    def _select(self, result_name, fixed_params):
        items = self[result_name]
        if fixed_params:
            keep = set(int(i) for i in np.atleast_1d(self.params.get_pack_indexes(fixed_params)))
            items = [v for i, v in enumerate(items) if i in keep]
        return items

    def get_result_values_list(self, result_name, fixed_params=None):
        return [v.get_result() for v in self._select(result_name, fixed_params)]

    def get_result_values_confidence_intervals(self, result_name, P=95.0, fixed_params=None):
        return [v.get_confidence_interval(P) for v in self._select(result_name, fixed_params)]

    # ---- persistence ------------------------------------------------------------------------
    def get_filename_with_replaced_params(self, filename):
        try:
            return replace_dict_values(filename, self.params.parameters, filename_mode=True)
        except (KeyError, IndexError, ValueError):
            return filename

    def to_dict(self):
        return {"params": self._params.to_dict(), "runned_reps": self.runned_reps,
                "original_filename": self.original_filename,
                "results": {n: [r.to_dict() for r in v] for n, v in self._results.items()}}

    @staticmethod
    def from_dict(d):
        sr = SimulationResults()
        sr._params = SimulationParameters.from_dict(d["params"])
        sr.runned_reps = d["runned_reps"]
        sr.original_filename = d["original_filename"]
        sr._results = {n: [Result.from_dict(r) for r in v] for n, v in d["results"].items()}
        return sr

    # the reference's (private) names, simulations/results.py:1361,1408 -- subclasses written against pyphysim call them
    _to_dict = to_dict
    _from_dict = from_dict

    def to_json(self):
        return json.dumps(self.to_dict(), default=_json_default)

    @staticmethod
    def from_json(text):
        return SimulationResults.from_dict(json.loads(text, object_hook=_json_object_hook))

    def save_to_file(self, filename):
        ext = os.path.splitext(filename)[-1]
        if ext == "":
            filename, ext = filename + ".pickle", ".pickle"
        if ext not in (".pickle", ".json"):
            raise KeyError(ext)
        self.original_filename = filename
        filename = self.get_filename_with_replaced_params(filename)
        if ext == ".pickle":
            with open(filename, "wb") as fh:
                pickle.dump(self, fh, protocol=2)
        else:
            with open(filename, "w") as fh:
                fh.write(self.to_json())
        return filename

    @staticmethod
    def load_from_file(filename):
        ext = os.path.splitext(filename)[-1]
        if ext == "":
            filename, ext = filename + ".pickle", ".pickle"
        if ext == ".pickle":
            with open(filename, "rb") as fh:
                obj = pickle.load(fh)
            assert isinstance(obj, SimulationResults)
            return obj
        if ext == ".json":
            with open(filename, "r") as fh:
                return SimulationResults.from_json(fh.read())
        raise KeyError(ext)

    def to_dataframe(self):
        import pandas as pd
        rows = self.params.get_unpacked_params_list()
        data = {name: [r[name] for r in rows] for name in self.params}
        for res in self:
            data[res[0].name] = [r.get_result() for r in res]
        if self.runned_reps is not None:
            data["runned_reps"] = self.runned_reps
        return pd.DataFrame(data)


def combine_simulation_results(simresults1, simresults2):
    """Merge two SimulationResults whose parameters differ only in the values of the unpacked ones: Results
    of variations present in both are merged, the others carried over (reference simulations/results.py:51-122;
    the bin/combine_results.py workflow)."""
    combined_params = combine_simulation_parameters(simresults1.params, simresults2.params)
    result_names = simresults1.get_result_names()
    if set(result_names) != set(simresults2.get_result_names()):
        raise RuntimeError("Both SimulationResults objects must have the same results.")
    union = SimulationResults()
    union.set_parameters(combined_params)
    for name in result_names:
        list1, list2 = simresults1[name], simresults2[name]
        type_code = list1[0].type_code
        for unpack in combined_params.get_unpacked_params_list():
            result_object = Result(name, type_code)
            fixed_parameters = unpack.parameters
            for sim, lst in ((simresults1, list1), (simresults2, list2)):
                try:
                    index = sim.params.get_pack_indexes(fixed_parameters)
                    result_object.merge(lst[index[0]])
                except ValueError:
                    pass
            union.append_result(result_object)
    return union
