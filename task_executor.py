"""Drop-in module name of the reference: re-exports artstyletransfer_amd.task_executor (MI355X HIP engine)."""
from artstyletransfer_amd import task_executor as _impl
from artstyletransfer_amd.task_executor import *  # noqa: F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
