"""Built-in environments (CPU env API of the reference; the device-resident synthetic env lives in device/)."""
from . import cartpole, grid, pendulum, synthetic_atari  # noqa: F401
