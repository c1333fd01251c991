"""2048_q-learning_amd -- MI355X-native batched 2048 tabular Q-learning hot path.

Drop-in for the path QLearningBase/environment/Game2048_env.py (step/reset) +
QLearningBase/Agent/main.py (choose_action/update_q_value) of Rocco9999/2048_Q-Learning,
running as hand-written HIP kernels for gfx950 behind the C ABI of include/q2048.h.

The directory name is not a Python identifier; import it with
    importlib.import_module("2048_q-learning_amd")
or through the repo-root shim `q2048_amd.py`.  The package needs the in-tree
csrc/libq2048_hip.so (built by __graft_entry__.build()).  Nothing is ever substituted for it: the
device "cpu" -- csrc/libq2048_host.so, the same C ABI compiled for the host from the kernels' own
per-lane arithmetic -- exists only for callers that ask for it by name.
"""
from . import _native
from ._native import NativeError, build
from .agent import (EPISODE_DTYPE, BatchedQLearningAgent, BatchedRowTupleAgent, EpisodeLog,
                    EpsilonSchedule, QLearningAgent, auto_capacity_log2, place_table,
                    stats_dict)
from . import launch
from .dist import Shard, StatsAllReduce, allreduce_stats, shard_plan, weak_shard
from .summary import SUMMARY_HEADER, summarize_csv, summarize_episodes, summarize_records, write_summary
from .env import (AUX_DTYPE, BatchedGame2048Env, Game2048_env, boards_to_raw, raw_to_boards)

__all__ = [
    "BatchedGame2048Env", "Game2048_env", "BatchedQLearningAgent", "BatchedRowTupleAgent",
    "QLearningAgent",
    "EpsilonSchedule", "EpisodeLog", "EPISODE_DTYPE", "stats_dict", "Shard", "shard_plan", "weak_shard", "allreduce_stats", "StatsAllReduce", "launch",
    "boards_to_raw", "raw_to_boards", "AUX_DTYPE", "build", "NativeError", "place_table", "auto_capacity_log2",
    "SUMMARY_HEADER", "summarize_csv", "summarize_episodes", "summarize_records", "write_summary",
]
