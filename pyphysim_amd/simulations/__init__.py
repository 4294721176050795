"""Monte Carlo driver surface of pyphysim.simulations kept for drop-in simulators:
SimulationRunner / SimulationResults / Result / SimulationParameters / SkipThisOne
(reference simulations/runner.py, results.py, parameters.py), plus the batched runner that
feeds whole batches of realizations from the GPU pipelines into the same accumulators."""
from .parameters import SimulationParameters, combine_simulation_parameters  # noqa: F401
from .results import Result, SimulationResults, calc_confidence_interval, combine_simulation_results  # noqa: F401
from .runner import (BatchedSimulationRunner, SimulationRunner, SkipThisOne,  # noqa: F401
                     get_partial_results_filename)
