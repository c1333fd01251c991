#!/usr/bin/env python3
"""Measurement tool (host only): why is the oracle's multi-threaded rollout (bench.py's cpu_baseline,
orc_rollout_mt: T threads x private Q-tables) slower on all 256 hardware threads of the GPU box than
on 64?  Varies T and who first touches a thread's table (the main thread through `reserve`, as
cpu_baseline does, or the worker itself by growing it), and reports the machine's topology."""
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402

O.lib()
cores = len(os.sched_getaffinity(0))
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
topo = {"usable_cpus": cores, "numa_nodes": len(nodes)}
try:
    sib = open("/sys/devices/system/cpu/cpu0/topology/thread_siblings_list").read().strip()
    topo["cpu0_thread_siblings"] = sib
    topo["physical_cores_guess"] = cores // max(1, len(sib.replace("-", ",").split(",")))
except OSError:
    pass
print(json.dumps({"topology": topo}), flush=True)
for T in (16, 32, 64, 128, 256):
    if T > cores:
        continue
    for touch in ("main", "worker"):
        B, steps = 16384 * T, 24
        envs = O.envs_init(B, 4, 0, 0)
        agents = [O.Agent(1000, 4, 0.1, 0.99, 0.95) for _ in range(T)]
        if touch == "main":
            for a in agents:
                a.reserve(16384 * 64)
        O.rollout_mt(envs, agents, 8, 0, 0, 0)
        t0 = time.perf_counter()
        O.rollout_mt(envs, agents, steps, 0, 0, 8)
        dt = time.perf_counter() - t0
        print(json.dumps({"threads": T, "first_touch": touch, "boards": B, "steps": steps,
                          "env_steps_per_s": B * steps / dt, "per_thread": B * steps / dt / T,
                          "seconds": round(dt, 2)}), flush=True)
        del agents, envs
