"""Flags: the counterpart of the reference's src/common/simple_arg_parse.py:17-34 and
src/common/config.py:17.  The reference scans sys.argv once at import for ``--flag=value``
pairs and casts to the type of the default; the same lookup is offered here, but nothing in
this package depends on argv implicitly -- constructors take explicit keyword arguments and
use these helpers only for their defaults, as the reference's SimulatedNetworkEnv does
(src/gym/network_sim.py:347-351)."""
import sys


def _scan(argv):
    table = {}
    for arg in argv:
        key, eq, value = arg.partition("=")
        table[key] = value if eq else True
    return table


_ARGS = _scan(sys.argv)


def arg_or_default(arg, default=None):
    """Value of ``arg`` on the command line (cast like the default), else ``default``."""
    if arg not in _ARGS:
        return default
    value = _ARGS[arg]
    if isinstance(default, bool):
        return value
    if isinstance(default, int):
        return int(value)
    if isinstance(default, float):
        return float(value)
    return value


DELTA_SCALE = arg_or_default("--delta-scale", 0.025)
