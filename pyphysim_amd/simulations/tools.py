"""Command-line workflows on result archives, mirroring the reference's bin/ scripts:

    python -m pyphysim_amd.simulations.tools combine FIRST SECOND [OUTPUT]     (bin/combine_results.py)
    python -m pyphysim_amd.simulations.tools split NAME [FOLDER]               (bin/split_into_partial_results.py)

Archives are the pickle / JSON files of SimulationResults.save_to_file; pickles written by pyphysim itself are
read through simulations.compat.
"""
import argparse
import os
import sys

from .parameters import replace_dict_values
from .results import SimulationResults, combine_simulation_results
from .runner import get_partial_results_filename


def _load(name):
    if os.path.splitext(name)[-1] in ("", ".pickle"):
        from . import compat
        return compat.load_reference_results(name if os.path.splitext(name)[-1] else name + ".pickle")
    return SimulationResults.load_from_file(name)


def combine_main(argv=None):
    """Combine two SimulationResults files into a new one (reference bin/combine_results.py:14-54)."""
    parser = argparse.ArgumentParser(prog="combine")
    parser.add_argument("first", help="The name of the first SimulationResults file.")
    parser.add_argument("second", help="The name of the second SimulationResults file.")
    parser.add_argument("output", nargs="?",
                        help="The name that will be used to save the combined SimulationResults file.")
    args = parser.parse_args(argv)
    first, second = _load(args.first), _load(args.second)
    union = combine_simulation_results(first, second)
    if args.output is None:
        output = replace_dict_values(first.original_filename, union.params.parameters, filename_mode=True)
    else:
        output = args.output
    if output in (args.first, args.second):
        raise RuntimeError("output filename must be different from the filename of either of the two "
                           "SimulationResults.")
    return union.save_to_file(output)


def split_main(argv=None):
    """Write one partial-result file per parameter variation of a SimulationResults file, named as the runner
    names them, so an interrupted sweep can be resumed from a finished archive (reference
    bin/split_into_partial_results.py:16-84)."""
    parser = argparse.ArgumentParser(prog="split")
    parser.add_argument("name", help="The name of the SimulationResults file.")
    parser.add_argument("folder", nargs="?", help="Folder for the partial result files.")
    args = parser.parse_args(argv)
    results = _load(args.name)
    original_filename = results.original_filename
    no_ext, ext = os.path.splitext(original_filename)
    if ext == ".pickle":
        original_filename = no_ext
    written = []
    unpacked = results.params.get_unpacked_params_list()
    names = results.get_result_names()
    for i, p in enumerate(unpacked):
        partial_filename = get_partial_results_filename(original_filename, p, args.folder)
        partial_filename = replace_dict_values(partial_filename, results.params.parameters, filename_mode=True)
        if partial_filename == args.name:
            raise RuntimeError("invalid name")
        partial = SimulationResults()
        partial.set_parameters(p)
        for n in names:
            partial.add_result(results[n][i])
        partial.current_rep = results.runned_reps[i]
        if args.folder is not None:
            os.makedirs(args.folder, exist_ok=True)
        written.append(partial.save_to_file(partial_filename))
    return written


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("combine", "split"):
        raise SystemExit(__doc__)
    out = combine_main(argv[1:]) if argv[0] == "combine" else split_main(argv[1:])
    print(out if isinstance(out, str) else "\n".join(out))


if __name__ == "__main__":
    main()
