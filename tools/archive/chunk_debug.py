#!/usr/bin/env python3
"""Debug: repeated q2048_table_alloc / free cycles with 4x4 and 5x5 rollouts; rows counted == rows inserted?

    python tools/archive/chunk_debug.py [chunks|plain]                 the shipped free path (the range stays reserved)
    Q2048_DEBUG_VA_FREE=1|2|3 python tools/archive/chunk_debug.py      the measurement build with round 3's free path:
        the address range is handed back (hipMemAddressFree) and the runtime may hand it out again --
        1: as round 3 did it, 2: + a device synchronize after the free, 3: + return codes on stderr.
    A table that does not arrive as zeros is now refused by q2048_table_alloc itself (Q2048_ERR_VERIFY):
    the tool reports that instead of running on it."""
import gc, importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("2048_q-learning_amd")
dev = "cuda:0"
placement = sys.argv[1] if len(sys.argv) > 1 else "chunks"
if os.environ.get("Q2048_DEBUG_VA_FREE"):
    pkg._native.use_experiments_build()
for it in range(6):
    n = 5 if it % 2 else 4
    env = pkg.BatchedGame2048Env(1 << 20, board_size=n, seed=31, env_id0=11, device=dev)
    try:
        agent = pkg.BatchedQLearningAgent(1000, learning_rate=0.1, discount_factor=0.99, exploration_rate=0.2,
                                          capacity_log2=27, seed=31, env_id0=11, device=dev, board_size=n,
                                          independent=True, placement=placement)
    except pkg.NativeError as exc:
        print(json.dumps({"it": it, "n": n, "alloc_refused": str(exc)}), flush=True)
        continue
    ptr = agent.table.data_ptr()
    nz0 = int((agent.table.view(torch.int64) != 0).sum())
    for _ in range(3):
        agent.fused_rollout(env, 16)
    torch.cuda.synchronize()
    st = agent.stats()
    q, found = agent.q_values(env.boards[:4096], return_found=True)
    print(json.dumps({"it": it, "n": n, "placement": agent.placement["mode"], "ptr": hex(ptr), "nonzero_words_on_arrival": nz0,
                      "inserts": st["inserts"], "table_size": agent.table_size(), "drops": st["drops"],
                      "current_states_found": int(found.sum()),
                      "claim_timeouts": pkg._native.claim_timeouts(), "status": agent.check_status()}), flush=True)
    del agent, env, q, found
    gc.collect()
