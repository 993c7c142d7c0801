"""Built-in environments (CPU env API of the reference; the device-resident synthetic env lives in device/)."""
from . import grid  # noqa: F401
