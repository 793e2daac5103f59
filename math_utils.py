"""Drop-in module name of the reference: re-exports artstyletransfer_amd.math_utils (MI355X HIP engine)."""
from artstyletransfer_amd import math_utils as _impl
from artstyletransfer_amd.math_utils import *  # noqa: F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
