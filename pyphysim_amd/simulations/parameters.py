"""SimulationParameters: a dict of parameters in which some iterable entries are marked to be
"unpacked" into the cartesian product of their values (reference simulations/parameters.py).
Host-only bookkeeping; nothing here touches the GPU."""
import copy
import itertools
import json
import pickle
from collections.abc import Iterable

import numpy as np


class SimulationParameters:
    def __init__(self):
        self.parameters = {}
        self._unpacked_parameters_set = set()
        self._unpack_index = -1
        self._original_sim_params = None

    # ---- construction -----------------------------------------------------------------------
    @staticmethod
    def _create(params_dict, unpack_index=-1, original_sim_params=None):
        sp = SimulationParameters()
        sp.parameters = copy.deepcopy(params_dict)
        sp._unpack_index = unpack_index if unpack_index >= 0 else -1
        sp._original_sim_params = original_sim_params
        return sp

    @staticmethod
    def create(params_dict):
        return SimulationParameters._create(params_dict)

    def add(self, name, value):
        self.parameters[name] = value

    def remove(self, name):
        del self.parameters[name]
        self._unpacked_parameters_set.discard(name)

    def set_unpack_parameter(self, name, unpack_bool=True):
        """parameters.py:327-358."""
        if name not in self.parameters:
            raise ValueError("Unknown parameter: `{0}`".format(name))
        if not isinstance(self.parameters[name], Iterable):
            raise ValueError("Parameter {0} is not iterable".format(name))
        if unpack_bool:
            self._unpacked_parameters_set.add(name)
        else:
            self._unpacked_parameters_set.remove(name)

    # ---- dict-like surface ------------------------------------------------------------------
    unpack_index = property(lambda self: self._unpack_index)
    unpacked_parameters = property(lambda self: sorted(self._unpacked_parameters_set))

    @property
    def fixed_parameters(self):
        return [n for n in self.parameters if n not in self._unpacked_parameters_set]

    def __getitem__(self, name):
        return self.parameters[name]

    def __setitem__(self, key, value):
        self.parameters[key] = value

    def __contains__(self, name):
        return name in self.parameters

    def __len__(self):
        return len(self.parameters)

    def __iter__(self):
        return iter(self.parameters)

    def __repr__(self):
        items = ["'{0}{1}': {2}".format(n, "*" if n in self._unpacked_parameters_set else "", v)
                 for n, v in self.parameters.items()]
        return "{%s}" % ", ".join(items)

    def __eq__(self, other):
        if self is other:
            return True
        if not isinstance(other, SimulationParameters):
            return False
        if (self._unpacked_parameters_set != other._unpacked_parameters_set
                or set(self.parameters) != set(other.parameters) or self._unpack_index != other._unpack_index):
            return False
        # 'rep_max' is allowed to differ (resuming a simulation with more repetitions)
        return not any(np.any(self.parameters[k] != other.parameters[k]) for k in self.parameters if k != "rep_max")

    def __ne__(self, other):
        return not self.__eq__(other)

    # ---- unpacking --------------------------------------------------------------------------
    def get_num_unpacked_variations(self):
        if self._original_sim_params is not None:
            return self._original_sim_params.get_num_unpacked_variations()
        n = 1
        for name in self._unpacked_parameters_set:
            n *= len(self.parameters[name])
        return n

    def get_unpacked_params_list(self):
        """One SimulationParameters per combination; unpacked names vary in SORTED-name order with
        the last name fastest (parameters.py:654-754)."""
        if not self._unpacked_parameters_set:
            return [self]
        names = sorted(self._unpacked_parameters_set)
        fixed = {k: v for k, v in self.parameters.items() if k not in self._unpacked_parameters_set}
        out = []
        for i, combo in enumerate(itertools.product(*[list(self.parameters[n]) for n in names])):
            d = dict(zip(names, combo))
            d.update(fixed)
            out.append(SimulationParameters._create(d, i, self))
        return out

    def get_pack_indexes(self, fixed_params_dict=None):
        """Indexes (into get_unpacked_params_list) of the variations whose unpacked parameters equal
        the given fixed values (parameters.py:576-652)."""
        fixed_params_dict = fixed_params_dict or {}
        names = self.unpacked_parameters
        dims = [len(self.parameters[n]) for n in names]
        grid = np.arange(self.get_num_unpacked_variations()).reshape(dims)
        sel = []
        for n in names:
            if n in fixed_params_dict:
                sel.append(list(self.parameters[n]).index(fixed_params_dict[n]))
            else:
                sel.append(slice(None))
        return np.asarray(grid[tuple(sel)]).flatten()

    # ---- persistence ------------------------------------------------------------------------
    def save_to_pickled_file(self, filename):
        with open(filename, "wb") as fh:
            pickle.dump(self, fh, protocol=2)

    @staticmethod
    def load_from_pickled_file(filename):
        with open(filename, "rb") as fh:
            return pickle.load(fh)

    def to_dict(self):
        orig = None if self._original_sim_params is None else self._original_sim_params.to_dict()
        return {"parameters": self.parameters, "unpacked_parameters_set": set(self._unpacked_parameters_set),
                "unpack_index": self._unpack_index, "original_sim_params": orig}

    @staticmethod
    def from_dict(d):
        sp = SimulationParameters()
        sp.parameters = {k: (np.asarray(v) if isinstance(v, list) else v) for k, v in d["parameters"].items()}
        sp._unpacked_parameters_set = set(d["unpacked_parameters_set"])
        sp._unpack_index = d["unpack_index"]
        if d.get("original_sim_params") is not None:
            sp._original_sim_params = SimulationParameters.from_dict(d["original_sim_params"])
        return sp

    # the reference's (private) names, simulations/parameters.py:942,963
    _to_dict = to_dict
    _from_dict = from_dict

    def to_json(self):
        return json.dumps(self.to_dict(), default=_json_default)

    @staticmethod
    def from_json(text):
        return SimulationParameters.from_dict(json.loads(text, object_hook=_json_object_hook))

    def to_dataframe(self):
        import pandas as pd
        rows = self.get_unpacked_params_list()
        return pd.DataFrame({name: [r[name] for r in rows] for name in self})


def get_range_representation(array, filename_mode=False):
    """'first:step:last' ('first_(step)_last' in file names) for an arithmetic progression of at least four
    values, a comma list for fewer, None otherwise (reference util/misc.py:911-960)."""
    array = np.asarray(array)
    if array.size < 4:
        return ",".join(array.astype(str))
    step = array[1] - array[0]
    if step.dtype == int:
        step = int(step)
    elif step.dtype == float:
        step = round(float(step), 12)
    if np.allclose(array[1:] - step, array[0:-1]):
        return ("{0}_({1})_{2}" if filename_mode else "{0}:{1}:{2}").format(array[0], step, array[-1])
    return None


def get_mixed_range_representation(array, filename_mode=False):
    """Comma-joined range representations of the constant-step stretches of `array`, cut where the step
    changes exactly as the reference cuts them (util/misc.py:963-1054) -- result file names depend on it."""
    array = np.asarray(array)
    if len(array) < 2:
        return "{0}".format(array[0])
    step_into = np.diff(array)
    step_into = np.concatenate([step_into[:1], step_into])     # step_into[i]: step that leads to element i
    pieces = []                                                  # [begin, end) index pairs
    begin, n = 0, len(step_into)
    while begin < n:
        end = begin
        while end < n and np.allclose(step_into[end], step_into[begin]):
            end += 1
        pieces.append([begin, end])
        begin = end
    # a stretch of more than three elements also claims the element before it when that one continues its step
    for i in range(1, len(pieces)):
        lo, hi = pieces[i]
        if hi - lo > 3:
            step = array[lo + 1] - array[lo]
            if np.allclose(array[lo] - array[pieces[i - 1][1] - 1], step):
                pieces[i - 1][1] -= 1
                pieces[i][0] -= 1
    out = []
    for lo, hi in pieces:
        text = get_range_representation(array[lo:hi], filename_mode)
        assert text is not None
        if text != "":
            out.append(text)
    return ",".join(out)


def replace_dict_values(name, dictionary, filename_mode=False):
    """`name.format(**dictionary)` with arrays shown as '[<mixed range representation>]'
    (reference util/misc.py:1057-1115)."""
    shown = {}
    for key, value in dictionary.items():
        if isinstance(value, np.ndarray):
            value = "[{0}]".format(get_mixed_range_representation(value, filename_mode))
        shown[key] = value
    return name.format(**shown)


def combine_simulation_parameters(params1, params2):
    """Union of two SimulationParameters that differ only in the VALUES of their unpacked parameters
    (reference simulations/parameters.py:55-107)."""
    if set(params1.parameters.keys()) != set(params2.parameters.keys()):
        raise RuntimeError("Both SimulationParameters objects must have the same parameters.")
    if set(params1.unpacked_parameters) != set(params2.unpacked_parameters):
        raise RuntimeError("Both SimulationParameters objects must have the same unpacked parameters (only the "
                           "values should can be different).")
    fixed = params1.fixed_parameters
    for key in fixed:
        if params1[key] != params2[key]:
            raise RuntimeError("The fixed parameters in both SimulationParameters objects must have the same value.")
    union = SimulationParameters()
    for key in fixed:
        union.add(key, copy.copy(params1[key]))
    for key in params1.unpacked_parameters:
        union.add(key, np.union1d(params1[key], params2[key]))
    for key in params1.unpacked_parameters:
        union.set_unpack_parameter(key)
    return union


def _json_default(obj):
    """The reference's NumpyOrSetEncoder (util/serialize.py:18-70): arrays and sets as tagged dicts, so JSON
    archives written here load in pyphysim and vice versa.  (NumPy floats are written as floats; the reference
    truncates them to int, :62-63.)"""
    if isinstance(obj, np.ndarray):
        return {"data": obj.tolist(), "dtype": str(obj.dtype), "_is_numpy_array": True, "shape": list(obj.shape)}
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, set):
        return {"data": sorted(obj), "_is_set": True}
    raise TypeError("not JSON serialisable: %r" % (type(obj),))


def _json_object_hook(dct):
    """json_numpy_or_set_obj_hook of the reference (util/serialize.py:73-110)."""
    if "_is_numpy_array" in dct:
        if dct["_is_numpy_array"] is True:
            return np.array(dct["data"], dtype=dct["dtype"]).reshape(dct["shape"])
        raise ValueError('Json representation contains the "_is_numpy_array" key but its value is not True')
    if "_is_set" in dct:
        if dct["_is_set"] is True:
            return set(dct["data"])
        raise ValueError('Json representation contains the "_is_set" key but its value is not True')
    return dct
