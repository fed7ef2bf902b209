"""dagl_amd: MI355X-native dynamic patch-graph attention block (DAGL ``CE``).

``from dagl_amd.ce import CE`` is the drop-in module; ``dagl_amd.ops`` wraps the C ABI of
``include/dagl_ce.h`` stage by stage; ``dagl_amd.build.build()`` compiles the HIP library in-tree.
"""
from ._lib import DaglError  # noqa: F401

__version__ = "0.1.0"
