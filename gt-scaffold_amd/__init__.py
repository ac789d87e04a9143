"""gt-scaffold hot path (scaffold-graph build, repeat marking, filtering, cycle
removal, scaffold construction) as an MI355X-native engine.

  csrc/      HIP kernels (gfx950) + the C-ABI (include/gt_scaffold_hip.h) + the
             C host layer that mirrors the reference's GtScaffolderGraph API
  engine.py  ctypes binding of the C-ABI (device memory handed over as raw
             pointers; torch is only used by callers for allocation/streams)
  synth.py   synthetic contig / DistEst / A-stat inputs
  dist.py    multi-GPU: component-partition step + sharded pipeline
"""
from . import synth  # noqa: F401
from . import dist  # noqa: F401

try:
    from . import engine  # noqa: F401
    from . import selftest  # noqa: F401
except ImportError:  # pragma: no cover - during early bring-up only
    engine = None
    selftest = None
