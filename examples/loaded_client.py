"""Drop-in for the reference's src/udt-plugins/testing/loaded_client.py: the file PCC-Uspace's Python rate controller loads by
name on the sender, e.g.

    ./pccclient send HOST PORT --pcc-rate-control=python -pyhelper=loaded_client -pypath=/path/to/this/repo/examples \
        --history-len=10 --pcc-utility-calc=linear --model-path=/path/to/exported/policy

It puts the repository on sys.path (like the reference file does for its own tree) and exposes the four functions the C++
side calls -- init(flow_id), get_rate(flow_id), give_sample(flow_id, ...), reset(flow_id) -- from pcc_rl_amd.udt_plugin.
No GPU is involved on the sender."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from pcc_rl_amd.udt_plugin import get_rate, give_sample, init, reset, set_policy  # noqa: E402,F401
