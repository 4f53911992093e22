"""`core` package mirror: re-exports the CorePyExt names exactly as the reference's core/__init__.py:2-4
does, from the in-tree MI355X build of the module."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)           # the reference locates CorePyExt through sys.path too (core/bin/__init__.py:20-25)

from . import lib as _lib  # noqa: E402
_lib.load()                             # brings torch's HIP runtime up first when torch is present (see lib._torch_first)

try:
    from CorePyExt import GameConfig, Player, Position, Board  # noqa: E402,F401
    from CorePyExt import Node, Policy, MCTS  # noqa: E402,F401
    from CorePyExt import RandomPolicy, PoolRAVEPolicy, TraditionalPolicy  # noqa: E402,F401
    from CorePyExt import set_seed, set_root_noise  # noqa: E402,F401  (extensions: reproducible searches, AddNoise parameters)
except ImportError as exc:              # fail loudly: there is no pure-Python stand-in
    raise ImportError("CorePyExt (MI355X build) is not built: run `python -m gomokuai_amd.build`") from exc

module_path = _HERE
