"""sdflib_amd — MI355X-native engine for the OctreeSdf / ExactOctreeSdf hot path of UPC-ViRVIG/SdfLib.

Hand-written HIP (gfx950) behind a C ABI (include/sdfhip.h, libsdfhip.so); this package is the thin host-side
mirror of the reference's interface.  No CPU fallback exists: importing works anywhere, but every compute entry
point requires the built library and a HIP device.
"""
from . import meshgen  # noqa: F401
from ._lib import SdfHipError, lib, LIB_PATH  # noqa: F401
from .api import (Context, Mesh, OctreeSdf, OctreeShard, ExactShard, ExactOctreeSdf, default_context, tricubic_fit, load_from_file,  # noqa: F401
                  string_to_termination_rule, HOST, DEVICE, RULE_NONE, RULE_TRAPEZOIDAL, RULE_SIMPSONS, RULE_BY_DISTANCE,
                  ALG_UNIFORM, ALG_NO_CONTINUITY, ALG_CONTINUITY, LAYOUT_GLOBAL_DFS, LAYOUT_SUBTREES, EVAL_EXACT, EVAL_FAST,
                  FIT_EXACT, FIT_MFMA)
