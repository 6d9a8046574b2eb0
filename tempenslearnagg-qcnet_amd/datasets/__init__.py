"""Input format of the path: the reference's processed-scenario files."""
from .processed_scenarios import (ProcessedScenarioDataset, UnsafeScenarioFile, collate_scenarios, load_scenario,  # noqa: F401
                                  save_scenario)
from .loader import scenario_loader  # noqa: F401
