"""wtp_amd — MI355X-native neighbour/stencil engine for WhatsThePoint.jl point clouds.

Host-side mirror of the reference's hot-path API (PointCloud / set_topology / repel) over the
C ABI of include/wtp.h.  All computation happens in csrc/libwtp.so on a gfx950 GPU; importing
this package never compiles or loads anything — the first Context() does, and fails loudly if
the library or the device is missing."""
from ._lib import WtpArgumentError, WtpError, build as build_library, load as load_library, SO_PATH
from .engine import Context, RelaxSession, default_context
from .topology import (AbstractTopology, NoTopology, KNNTopology, RadiusTopology, CSR, isvalid)
from .cloud import (PointSurface, PointVolume, PointBoundary, PointCloud, points, set_topology,
                    rebuild_topology, hastopology, neighbors, KNearestSearch, search, searchdists)
from .forces import (RepelForceModel, InverseDistanceForce, SpacingEquilibriumForce, ClippedSpacingForce,
                     StrongSpacingForce, compute_force)
from .spacings import ConstantSpacing, LogLike, BoundaryLayerSpacing
from .repel import repel, relax
from .metrics import metrics, spacing_metrics, spacing_fidelity_metrics
from .inside import isinside
from .octree import TriangleOctree, has_consistent_normals, signed_volume
from .normals import compute_normals, update_normals, orient_normals, split_surface, combine_surfaces
from .limiter import gradient_limit_field
from . import synth, stl, octree

__all__ = [n for n in dir() if not n.startswith("_")]
