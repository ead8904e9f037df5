"""polytope_amd -- MI355X (gfx950) engine for the batched small-LP hot path of
tulip-control/polytope: Chebyshev-ball / redundancy / bounding-box LPs behind `reduce`,
`intersect`, `region_diff`; dense containment behind `contains`/`is_inside`; the
distance / furthest-point kernels of `quickhull`.

The compute path is hand-written HIP (polytope_amd/csrc, built into libplp_hip.so and
reached through the C ABI of include/plp.h).  There is no CPU fallback: without the
library or without a gfx950 device the 'hip' backend raises.
"""
from . import _lib  # noqa: F401
from .batch import (  # noqa: F401
    lpsolve_batch, cheby_ball_batch, bbox_batch, reduce_batch, contains_batch, assign_batch, adjacent_pairs, keep_to_bool,
)

__version__ = "0.1.0"
