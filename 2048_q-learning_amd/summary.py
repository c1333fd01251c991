"""Run summaries in the layout of the reference's published aggregate,
QLearningBase/plots/summary_statistics_cleaned.csv (header, line 1):

    Reward_Technique,Avg_Reward,Std_Reward,Max_Value,Action_0,Action_1,Action_2,Action_3

one row per training log: mean and standard deviation of the per-episode `Reward` column (the
reward of the episode's last step, Agent/main.py:62,105), the largest `Max Value`, and how often
each action ended an episode.  The reference ships the table but not the script that made it;
the standard deviation here is the sample one (ddof = 1, what pandas' `.std()` gives).  Inputs:
the per-episode CSV of Agent/main.py:71-76 (as written by the reference or by `train.py`) or
the records of a device-side `EpisodeLog`.  Host-side arithmetic on a few columns: no GPU."""
from __future__ import annotations

import csv

import numpy as np

SUMMARY_HEADER = ["Reward_Technique", "Avg_Reward", "Std_Reward", "Max_Value",
                  "Action_0", "Action_1", "Action_2", "Action_3"]


def summarize_episodes(name: str, actions, rewards, max_values, ddof: int = 1) -> list:
    """One summary row from per-episode columns (any array-likes of equal length)."""
    actions = np.asarray(actions, dtype=np.int64).reshape(-1)
    rewards = np.asarray(rewards, dtype=np.float64).reshape(-1)
    max_values = np.asarray(max_values, dtype=np.int64).reshape(-1)
    if not (len(actions) == len(rewards) == len(max_values)) or len(actions) == 0:
        raise ValueError("need equally long, non-empty action / reward / max-value columns")
    if actions.min() < 0 or actions.max() > 3:
        raise ValueError("actions must be 0..3")
    counts = np.bincount(actions, minlength=4)
    std = float(rewards.std(ddof=ddof)) if len(rewards) > ddof else float("nan")
    return [name, float(rewards.mean()), std, int(max_values.max())] + [int(c) for c in counts]


def summarize_records(name: str, records: np.ndarray, ddof: int = 1) -> list:
    """Summary row of `EpisodeLog.drain()` records (EPISODE_DTYPE)."""
    return summarize_episodes(name, records["action"], records["reward"],
                              np.left_shift(1, records["max_log2"].astype(np.int64)), ddof)


def summarize_csv(path: str, name: str | None = None, ddof: int = 1) -> list:
    """Summary row of a per-episode CSV with the reference's columns
    Episode,Action,Q-Values,Reward,Total-Reward,Max Value (extra columns are ignored)."""
    actions, rewards, max_values = [], [], []
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            actions.append(int(row["Action"]))
            rewards.append(float(row["Reward"]))
            max_values.append(int(float(row["Max Value"])))
    if name is None:
        name = path.rsplit("/", 1)[-1]
        name = name[:-4] if name.endswith(".csv") else name
    return summarize_episodes(name, actions, rewards, max_values, ddof)


def write_summary(rows, path: str) -> None:
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(SUMMARY_HEADER)
        w.writerows(rows)
