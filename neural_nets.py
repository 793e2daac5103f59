"""Drop-in module name of the reference: re-exports artstyletransfer_amd.neural_nets (MI355X HIP engine)."""
from artstyletransfer_amd import neural_nets as _impl
from artstyletransfer_amd.neural_nets import *  # noqa: F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
