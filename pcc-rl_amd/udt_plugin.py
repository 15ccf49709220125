"""The module PCC-Uspace's Python hook loads on the sender (`--pcc-rate-control=python -pyhelper=...`): four
module-level functions keyed by flow id -- init, get_rate, give_sample, reset -- exactly the surface of the
reference's src/udt-plugins/testing/loaded_client.py:132-173, with a policy exported by pcc_rl_amd.export behind
them instead of a TensorFlow SavedModel (loaded_agent.py).

Arguments come from sys.argv like the reference's (simple_arg_parse, loaded_client.py:44-52):
    --model-path=DIR          directory written by pcc_rl_amd.export.export_policy (policy.pt + signature.json)
    --history-len=10 --input-features="sent latency inflation,latency ratio,send ratio"
    --reset-target-rate=6.0   starting rate in Mbps (the reference's RESET_RATE_MIN = RESET_RATE_MAX = 6.0)

No GPU is involved here: this runs inside the sender process, one call per monitor interval.
"""
import sys

from .config import arg_or_default
from .metrics import DEFAULT_FEATURES
from .shim import PolicyRateController

if not hasattr(sys, "argv"):        # embedded interpreters may not set it (loaded_client.py:30-31)
    sys.argv = [""]

MIN_RATE, MAX_RATE, DELTA_SCALE = 0.5, 300.0, 0.05     # loaded_client.py:33-35
_flows = {}                                             # PccGymDriver.flow_lookup (loaded_client.py:58)
_act = None                                             # one loaded policy serves every flow


def set_policy(act):
    """Use `act(obs) -> action` instead of loading --model-path (tests; embedding a policy object)."""
    global _act
    _act = act


def _policy():
    global _act
    if _act is None:
        from .export import load_policy
        _act = load_policy(arg_or_default("--model-path", default="/tmp/"))
    return _act


def init(flow_id):
    """loaded_client.py:172-173 / 60-80: a driver for the flow, at the reset rate, with an empty history.  A flow id that
    is initialised AGAIN keeps its latency minimum: the reference's table of connection minima is module-level and keyed
    by flow id (so:158-176), a new PccGymDriver for the same id does not clear it -- so the empty intervals of the new
    history evaluate "latency ratio" to 0 / min = 0.0, not the 1.0 of a first history."""
    prev = _flows.get(flow_id)
    _flows[flow_id] = PolicyRateController(
        _policy(), history_len=arg_or_default("--history-len", default=10),
        features=arg_or_default("--input-features", default=DEFAULT_FEATURES),
        start_rate=float(arg_or_default("--reset-target-rate", default=6.0)),
        delta_scale=DELTA_SCALE, min_rate=MIN_RATE, max_rate=MAX_RATE,
        conn_min=prev.history.conn_min if prev is not None else None)


def get_rate(flow_id):
    """loaded_client.py:167-170: the rate in bits/s; the policy acts once a sample has arrived."""
    return _flows[flow_id].get_rate()


def give_sample(flow_id, bytes_sent, bytes_acked, bytes_lost, send_start_time, send_end_time, recv_start_time,
                recv_end_time, rtt_samples, packet_size, utility):
    """loaded_client.py:132-138: one finished monitor interval of the flow."""
    _flows[flow_id].give_sample(bytes_sent, bytes_acked, bytes_lost, send_start_time, send_end_time, recv_start_time,
                                recv_end_time, rtt_samples, packet_size, utility, flow_id=flow_id)


def reset(flow_id):
    """loaded_client.py:163-165: fresh rate and history for the flow (its latency minimum survives, like the
    reference's module-level table)."""
    _flows[flow_id].reset()
