"""Built-in algorithm plugins, registered under the reference's names ("QL", "DQN:torch", "Rainbow:torch",
"Rainbow_no_multisteps:torch", "Agent57_light:torch", "Agent57:torch") so that a config written for the reference resolves to these classes."""
from . import agent57, agent57_light, dqn, ql, rainbow  # noqa: F401
