"""Public names of the package (re-exported by the ``pcc_rl_amd`` import shim)."""
from .build import build_library, library_path
from .config import DELTA_SCALE, arg_or_default
from .env import BatchedNetworkEnv, GroupedNetworkEnv, SimulatedNetworkEnv, make, register_gym
from .metrics import (DEFAULT_FEATURES, METRIC_NAMES, feature_ids, get_max_obs_vector,
                      get_min_obs_vector, metric_info)
from .native import PccError, STEP_COLUMNS
from .spaces import Box

__all__ = ["BatchedNetworkEnv", "GroupedNetworkEnv", "SimulatedNetworkEnv", "make", "register_gym", "Box", "PccError",
           "build_library", "library_path", "DELTA_SCALE", "arg_or_default", "DEFAULT_FEATURES",
           "METRIC_NAMES", "STEP_COLUMNS", "feature_ids", "get_min_obs_vector", "get_max_obs_vector",
           "metric_info"]
