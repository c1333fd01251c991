#!/usr/bin/env python3
"""Measurement tool: where in ONE fused launch the chip is not full.  The measurement build stamps every
block's entry, first step, last step and exit with the 100 MHz wall clock (q2048_debug_timeline); this
prints, for the driver's 20-step launch at 1 Mi boards: when blocks start and end (the three "rounds" of
1536 resident blocks), how long a block's first step takes against its later ones, the number of blocks in
flight over time, and how much of the launch runs below 2/3 of full residency (ramp and drain)."""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
pkg._native.use_experiments_build()
L = pkg._native.lib()
L.q2048_debug_timeline.restype, L.q2048_debug_timeline.argtypes = C.c_int, [C.c_void_p]
dev = torch.device("cuda:0")
B, S = 1 << 20, int(os.environ.get("TIMELINE_STEPS", "20"))
env = pkg.BatchedGame2048Env(B, seed=0, device=dev)
agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.95,
                                  capacity_log2=30, seed=0, device=dev)
agent.fused_rollout(env, 512, play_only=True)
agent.ctr = env.ctr
agent.fused_rollout(env, 25)
BLOCK = 512 if B >= 786432 else 256                 # the library's choice (csrc: kFusedBigBatch)
blocks = B // BLOCK
stamps = torch.zeros((blocks, 8), dtype=torch.int64, device=dev)
for rep in range(3):
    stamps.zero_()
    assert L.q2048_debug_timeline(stamps.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); agent.fused_rollout(env, S); e1.record()
    torch.cuda.synchronize()
    assert L.q2048_debug_timeline(None) == 0
    raw = stamps.cpu().numpy()
    if not (raw[:, 0] > 0).all() or not (raw[:, 3] >= raw[:, 0]).all():   # a block that left no stamps: wrong grid assumed
        raise SystemExit(f"{int((raw[:, 0] == 0).sum())} of {blocks} blocks left no stamps (block size {BLOCK}?)")
    hw, xcc = raw[:, 4], raw[:, 5] & 15
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7        # gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    t = raw[:, :4].astype(np.float64) / 100.0                    # us
    t -= t[:, 0].min()
    start, first, last, end = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    total = end.max()
    if total > 1e6:
        raise SystemExit(f"a launch of {total:.0f} us? stamps are not from one launch")
    grid = np.arange(0.0, total, 1.0)
    inflight = np.array([((start <= x) & (end > x)).sum() for x in grid])
    full = inflight.max()
    order = np.argsort(start)
    resident = 1536 * 256 // BLOCK                   # 6 waves per SIMD
    rounds = [order[:resident], order[resident:2 * resident], order[2 * resident:]]
    out = {"steps": S, "launch_us_by_events": round(e0.elapsed_time(e1) * 1e3, 1), "span_us_by_stamps": round(total, 1),
           "max_blocks_in_flight": int(full),
           "us_below_two_thirds_of_full": {"at_the_start": int((inflight[: len(grid) // 2] < full * 2 / 3).sum()),
                                            "at_the_end": int((inflight[len(grid) // 2:] < full * 2 / 3).sum())},
           "rounds": [{"blocks": int(len(r)), "start_us": [round(float(start[r].min()), 1), round(float(np.median(start[r])), 1), round(float(start[r].max()), 1)],
                       "end_us": [round(float(end[r].min()), 1), round(float(np.median(end[r])), 1), round(float(end[r].max()), 1)],
                       "first_step_us_median": round(float(np.median((first - start)[r])), 1),
                       "later_step_us_median": round(float(np.median(((last - first) / max(S - 1, 1))[r])), 1),
                       "epilogue_us_median": round(float(np.median((end - last)[r])), 1)} for r in rounds if len(r)],
           "round_1_block_us_by_xcc": {int(x): [int((xcc[rounds[0]] == x).sum()), round(float(np.median((end - start)[rounds[0]][xcc[rounds[0]] == x])), 1)]
                                       for x in sorted(set(xcc.tolist()))},
           "blocks_run_by_xcc": {int(x): int((xcc == x).sum()) for x in sorted(set(xcc.tolist()))},
           "last_50_blocks_to_end": sorted({(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])) for b in np.argsort(end)[-50:]}),
           "round_1_block_us_percentiles": [round(float(np.percentile((end - start)[rounds[0]], q)), 1) for q in (1, 10, 50, 90, 99)],
           "round_1_slowest_cu_median_vs_fastest": (lambda d: [round(float(min(d.values())), 1), round(float(max(d.values())), 1), len(d)])(
               {key: float(np.median([(end - start)[b] for b in rounds[0] if (int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])) == key]))
                for key in {(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])) for b in rounds[0]}}),
           "blocks_in_flight_every_50us": [int(v) for v in inflight[::50]],
           "blocks_in_flight_last_60us": [int(v) for v in inflight[-60::5]]}
    print(json.dumps(out), flush=True)
