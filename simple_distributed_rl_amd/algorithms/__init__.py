"""Built-in algorithm plugins, registered under the reference's names ("QL", "DQN:torch", "Rainbow:torch",
"Rainbow_no_multisteps:torch", "Agent57_light:torch", "Agent57:torch"; "PPO:torch" stands in for the reference's
TensorFlow PPO) so that a config written for the reference resolves to these classes."""
from . import agent57, agent57_light, dqn, ppo, ql, rainbow  # noqa: F401
