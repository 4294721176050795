"""Reading and writing the reference's pickled result files.

pyphysim stores ``SimulationResults`` / ``SimulationParameters`` / ``Result`` objects with
``pickle`` (protocol 2; reference simulations/results.py:1454-1535, runner.py:926-994).  The
classes here keep the reference's attribute names on purpose, so its archives load by mapping the
class paths -- no reference code is imported -- and archives written here load in the reference.
"""
import io
import pickle

from . import parameters as _parameters
from . import results as _results

_CLASS_MAP = {
    ("pyphysim.simulations.results", "SimulationResults"): _results.SimulationResults,
    ("pyphysim.simulations.results", "Result"): _results.Result,
    ("pyphysim.simulations.parameters", "SimulationParameters"): _parameters.SimulationParameters,
}
_REVERSE = {v: k for k, v in _CLASS_MAP.items()}


class _ReferenceUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _CLASS_MAP:
            return _CLASS_MAP[(module, name)]
        if module.startswith("pyphysim"):
            raise pickle.UnpicklingError("unsupported reference class %s.%s" % (module, name))
        return super().find_class(module, name)


def load_reference_results(filename):
    """Load a ``SimulationResults`` pickle written by pyphysim (full or partial results file)."""
    with open(filename, "rb") as fh:
        try:
            obj = _ReferenceUnpickler(fh).load()
        except UnicodeDecodeError:          # Python-2 era archives (results.py:1546-1552)
            fh.seek(0)
            obj = _ReferenceUnpickler(fh, encoding="iso-8859-1").load()
    if not isinstance(obj, _results.SimulationResults):
        raise TypeError("%s does not hold a SimulationResults object" % filename)
    if not hasattr(obj, "current_rep"):
        obj.current_rep = -1
    return obj


class _ReferencePickler(pickle._Pickler):          # the pure-Python pickler: save_global can be redirected
    def save_global(self, obj, name=None):
        target = _REVERSE.get(obj)
        if target is None:
            return super().save_global(obj, name)
        module, cls_name = target
        self.write(pickle.GLOBAL + module.encode("ascii") + b"\n" + cls_name.encode("ascii") + b"\n")
        self.memoize(obj)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = save_global


def save_for_reference(results, filename):
    """Write `results` so that pyphysim's ``SimulationResults.load_from_file`` reads it back."""
    state = {k: v for k, v in results.__dict__.items() if k != "_batched_state"}
    clone = _results.SimulationResults.__new__(_results.SimulationResults)
    clone.__dict__.update(state)
    buf = io.BytesIO()
    _ReferencePickler(buf, protocol=2).dump(clone)
    with open(filename, "wb") as fh:
        fh.write(buf.getvalue())
    return filename
