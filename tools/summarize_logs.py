#!/usr/bin/env python3
"""Per-episode CSV logs -> one summary table in the reference's layout
(QLearningBase/plots/summary_statistics_cleaned.csv).
    python tools/summarize_logs.py debug_log.csv other_log.csv -o summary_statistics.csv"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("logs", nargs="+")
    p.add_argument("-o", "--out", default="summary_statistics.csv")
    args = p.parse_args()
    summary = importlib.import_module("2048_q-learning_amd.summary")
    rows = [summary.summarize_csv(path) for path in args.logs]
    summary.write_summary(rows, args.out)
    for r in rows:
        print(",".join(str(x) for x in r))


if __name__ == "__main__":
    main()
