"""Monitor-interval metric registry: names, bounds and scales of the reference's
src/common/sender_obs.py:193-206.  The metric *formulas* (so:110-191) run on the GPU inside
the step kernel; this module only carries the table the env surface needs to build its
observation space (get_min_obs_vector / get_max_obs_vector, so:95-108) and to turn the
reference's comma-separated feature string (src/gym/network_sim.py:348-351) into ids."""
import numpy as np

# (name, min, max, scale) in registry order; the position is the id the C ABI uses
_TABLE = [
    ("send rate", 0.0, 1e9, 1e7),
    ("recv rate", 0.0, 1e9, 1e7),
    ("recv dur", 0.0, 100.0, 1.0),
    ("send dur", 0.0, 100.0, 1.0),
    ("avg latency", 0.0, 100.0, 1.0),
    ("loss ratio", 0.0, 1.0, 1.0),
    ("ack latency inflation", -1.0, 10.0, 1.0),
    ("sent latency inflation", -1.0, 10.0, 1.0),
    ("conn min latency", 0.0, 100.0, 1.0),
    ("latency increase", 0.0, 100.0, 1.0),
    ("latency ratio", 1.0, 10000.0, 1.0),
    ("send ratio", 0.0, 1000.0, 1.0),
]
METRIC_NAMES = [row[0] for row in _TABLE]
DEFAULT_FEATURES = "sent latency inflation,latency ratio,send ratio"


def _names(features):
    if isinstance(features, str):
        features = features.split(",")
    return [f.strip() for f in features]


def feature_ids(features):
    ids = []
    for name in _names(features):
        if name not in METRIC_NAMES:
            raise KeyError("unknown monitor-interval metric %r (known: %s)" % (name, ", ".join(METRIC_NAMES)))
        ids.append(METRIC_NAMES.index(name))
    return ids


def metric_info(name):
    """(min_val, max_val, scale) of one metric."""
    row = _TABLE[METRIC_NAMES.index(name)]
    return row[1], row[2], row[3]


def get_min_obs_vector(features):
    return np.array([metric_info(n)[0] for n in _names(features)])


def get_max_obs_vector(features):
    return np.array([metric_info(n)[1] for n in _names(features)])
