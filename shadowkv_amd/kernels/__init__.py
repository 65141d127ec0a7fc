"""Mirror of the reference's `kernels` package: `from shadowkv_amd.kernels import shadowkv`."""
from . import shadowkv  # noqa: F401
