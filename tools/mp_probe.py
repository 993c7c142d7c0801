"""Probe: Runner.train_mp (2 actor processes + the device trainer) with DQN on CartPole-v1, everything sharing one GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import dqn
from simple_distributed_rl_amd.utils.common import set_seed

if __name__ == "__main__":
    set_seed(3, enable_gpu=True)
    rl = dqn.Config(batch_size=32, lr=0.001, target_model_update_interval=200, discount=0.99)
    rl.memory.set_replay_buffer()
    rl.memory.capacity, rl.memory.warmup_size = 100_000, 500
    rl.epsilon_scheduler.set_linear(1.0, 0.05, 3000)
    rl.hidden_block.set((64, 64))
    runner = srl.Runner("CartPole-v1", rl)
    runner.set_device("cuda:0")
    t0 = time.time()
    st = runner.train_mp(actor_num=2, max_train_count=4000, timeout=240, trainer_parameter_send_interval=0.5, actor_parameter_sync_interval=0.5, enable_progress=False)
    print("train_mp done in", round(time.time() - t0, 1), "s; train_count", st.train_count, st.end_reason, "recv", st.trainer_recv_q)
    r = runner.evaluate(max_episodes=10, enable_progress=False)
    print("eval", np.mean(r), r)
