"""polytope_amd -- MI355X (gfx950) engine for the batched small-LP hot path of
tulip-control/polytope: Chebyshev-ball / redundancy / bounding-box LPs behind `reduce`,
`intersect`, `region_diff`; dense containment behind `contains`/`is_inside`; the
distance / furthest-point kernels of `quickhull`.

Drop-in surface: the package root carries the names the reference's root carries
(polytope/__init__.py:35-44) for everything on that path, so

    import polytope_amd as polytope
    polytope.solvers.default_solver = 'hip'

is the whole switch.  `grid_region` and `projection` (and plotting) are outside the hot path
 and are not provided.  The batched entry points (`*_batch`, no reference
counterpart) sit beside them.

The compute path is hand-written HIP (polytope_amd/csrc, built into libplp_hip.so and
reached through the C ABI of include/plp.h).  There is no CPU fallback: without the
library or without a gfx950 device the 'hip' backend raises.
"""
from . import _lib  # noqa: F401
from . import solvers  # noqa: F401
from .polytope import (  # noqa: F401
    Polytope, Region,
    is_empty, is_fulldim, is_convex, is_adjacent, is_subset,
    reduce, separate, box2poly,
    cheby_ball, bounding_box, envelope, extreme, qhull,
    is_inside, union, mldivide, intersect, volume,
)
from .prop2partition import (  # noqa: F401
    Partition, MetricPartition, find_adjacent_regions)
from . import polytope, prop2partition, quickhull  # noqa: F401,E402  (submodules, as `polytope.polytope` etc.)
from .batch import (  # noqa: F401
    lpsolve_batch, cheby_ball_batch, bbox_batch, reduce_batch, contains_batch, assign_batch, adjacent_pairs, keep_to_bool,
    verify_careful_lps, lp_histograms,
)

__version__ = "0.1.0"
