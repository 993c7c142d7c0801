"""simple_distributed_rl_amd -- MI355X (gfx950) native actor-learner data path behind the
SRL (pocokhc/simple_distributed_rl) Runner / Config / Worker / Trainer / Memory plugin surface.

The compute path is hand-written HIP behind the C ABI of include/srlx.h (libsrlx.so); this
package is the Python host that mirrors the reference's plugin interfaces.  There is no CPU
fallback: using a device-backed class without the built library raises.
"""

__version__ = "0.1.0"

from simple_distributed_rl_amd.base.env.registration import EnvConfig  # noqa: E402,F401
from simple_distributed_rl_amd.base.env.registration import make as make_env  # noqa: E402,F401
from simple_distributed_rl_amd.runner.runner import Runner  # noqa: E402,F401
