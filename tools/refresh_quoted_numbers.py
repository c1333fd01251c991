#!/usr/bin/env python3
"""Re-writes the figures that README.md, DESIGN.md and INTEGRATION.md quote from this round's bench profiles
(profiles/r06_bench_k20.json, r06_bench.json, r06_four_call.jsonl, r06_deterministic_mode.jsonl, r06_adapters.json)
after those files have been replaced by a newer run, together with the phrases of profiles/quoted_figures.json that
tests/test_docs_quote_profiles.py checks.  Only figures that come from those files are touched; everything else in the
documents is prose.      python tools/refresh_quoted_numbers.py"""
import json
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(REPO, *a)   # noqa: E731


def jl(path):
    return [json.loads(ln) for ln in open(path) if ln.startswith("{")]


k20, df = json.load(open(P("profiles", "r06_bench_k20.json"))), json.load(open(P("profiles", "r06_bench.json")))
comp = lambda d, pre: [c for c in d["companions"] if c["name"].startswith(pre)][0]   # noqa: E731
v20, ms20, l20 = k20["value"] / 1e10, k20["ms_per_step"] * 1e3, k20["roofline"]["avg_launch_ms"]
f20, ff20 = k20["roofline"]["frac"], k20["roofline"]["fabric_frac"]
vd, msd, ld, fd, ffd = (df["value"] / 1e10, df["ms_per_step"] * 1e3, df["roofline"]["avg_launch_ms"], df["roofline"]["frac"],
                        df["roofline"]["fabric_frac"])
fz20, fzd, c5, c5d = comp(k20, "frozen"), comp(df, "frozen"), comp(k20, "5x5"), comp(df, "5x5")
fc1 = [r for r in jl(P("profiles", "r06_four_call.jsonl")) if r["B"] == 1048576 and r.get("board") == 4][0]["us_per_step"]
det1 = [r for r in jl(P("profiles", "r06_deterministic_mode.jsonl")) if r["mode"] == "deterministic" and r["B"] == 1048576][0]["us_per_step"]
ads = f"{json.load(open(P('profiles', 'r06_adapters.json')))['iterations_per_s']:,.0f}".replace(",", " ")
cb, pc = k20["cpu_baseline"], k20["cpu_baseline"]["product_core"]
NUM = r"[0-9]+(?:\.[0-9]+)?"
SUP = str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹")


def sci(x: float) -> str:
    """1.07·10⁸: one digit before the point."""
    e = len(str(int(x))) - 1
    return f"{x / 10 ** e:.1f}·10" + str(e).translate(SUP)


def sub(text, pattern, repl, what):
    out, n = re.subn(pattern, repl, text)
    if n != 1:
        raise SystemExit(f"{what}: pattern matched {n} times")
    return out


# ---- DESIGN.md ----------------------------------------------------------------------------------------------------
s = open(P("DESIGN.md"), encoding="utf-8").read()
s = sub(s, rf"\*\*{NUM}·10¹⁰ env-steps/s, {NUM} µs per step, launch {NUM} ms by HIP events = {NUM} of 8 TB/s, `fabric_frac` {NUM}\*\*",
        f"**{v20:.2f}·10¹⁰ env-steps/s, {ms20:.1f} µs per step, launch {l20:.3f} ms by HIP events = {f20:.3f} of 8 TB/s, `fabric_frac` {ff20:.3f}**",
        "DESIGN k20 line")
s = sub(s, rf"\*\*{NUM}·10¹⁰, {NUM} µs per step, {NUM} ms per launch = {NUM}, `fabric_frac` {NUM}\*\*",
        f"**{vd:.2f}·10¹⁰, {msd:.1f} µs per step, {ld:.3f} ms per launch = {fd:.3f}, `fabric_frac` {ffd:.3f}**", "DESIGN default line")
s = sub(s, rf"\*\*{NUM} µs per step, {NUM}·10¹⁰, frac {NUM}\*\* on the driver's command; {NUM} µs / {NUM} on 64-step launches \(bench companion",
        f"**{fz20['ms_per_step'] * 1e3:.1f} µs per step, {fz20['value'] / 1e10:.2f}·10¹⁰, frac {fz20['roofline_frac']:.3f}** on the driver's command; "
        f"{fzd['ms_per_step'] * 1e3:.1f} µs / {fzd['roofline_frac']:.3f} on 64-step launches (bench companion", "DESIGN frozen companion")
s = sub(s, rf"\| {NUM} µs, {NUM}·10¹⁰, frac {NUM} \(driver's command;", f"| {c5['ms_per_step'] * 1e3:.1f} µs, {c5['value'] / 1e10:.2f}·10¹⁰, frac {c5['roofline_frac']:.3f} (driver's command;",
        "DESIGN 5x5 k20")
s = sub(s, rf"0\.316\); {NUM} µs, {NUM}·10¹⁰, {NUM} \(64-step launches\)", f"0.316); {c5d['ms_per_step'] * 1e3:.1f} µs, {c5d['value'] / 1e10:.2f}·10¹⁰, {c5d['roofline_frac']:.3f} (64-step launches)",
        "DESIGN 5x5 default")
s = sub(s, rf"{NUM} µs per step \(`r06_four_call\.jsonl`", f"{fc1:.1f} µs per step (`r06_four_call.jsonl`", "DESIGN four-call")
s = sub(s, r"the one-env adapters: [0-9 ]+ loop iterations/s", f"the one-env adapters: {ads} loop iterations/s", "DESIGN adapters")
s = sub(s, rf"\| random reads \+ stable partition \| {NUM} µs per step \(`r06_deterministic_mode\.jsonl`\)",
        f"| random reads + stable partition | {det1:.1f} µs per step (`r06_deterministic_mode.jsonl`)", "DESIGN det")
s = sub(s, rf"That is 4\.0 GB per\n{NUM} ms launch", f"That is 4.0 GB per\n{l20:.3f} ms launch", "DESIGN launch in the traffic paragraph")
bt, ot = pc["by_threads"], cb["by_threads"]
s = sub(s, r"(?s)`cpu_baseline` on the GPU box's host \(2 × EPYC 9575F, `r06_bench_k20\.json`\):.*?(?=\n\n## 6\. Multi-GPU)",
        (f"`cpu_baseline` on the GPU box's host (2 × EPYC 9575F, `r06_bench_k20.json`): kind `\"port\"` — the oracle, a C port\n"
         f"of the reference loop: {sci(cb['value'])} env-steps/s on {cb['cores']} threads (private tables; {sci(ot['256'])} on all 256), "
         f"{sci(cb['single_thread']['value'])} on\none; and `product_core` — the CPU twin, one shared table like the GPU: **{sci(pc['value'])} on {pc['cores']} threads "
         f"({sci(bt['16'])} on 16,\n{sci(bt['256'])} on 256), {sci(pc['single_thread']['value'])} on one**. Reference CPython: 1.1·10⁴. The GPU line is "
         f"{k20['value'] / pc['value']:.0f} × the product's own code on\nthe host's best thread count."), "DESIGN cpu baseline")
open(P("DESIGN.md"), "w", encoding="utf-8").write(s)

# ---- README.md ----------------------------------------------------------------------------------------------------
s = open(P("README.md"), encoding="utf-8").read()
s = sub(s, rf"\*\*{NUM}·10¹⁰ env-steps/s, {NUM} µs per step, {NUM} of the 8 TB/s HBM roofline\*\*",
        f"**{v20:.2f}·10¹⁰ env-steps/s, {ms20:.1f} µs per step, {f20:.3f} of the 8 TB/s HBM roofline**", "README k20")
s = sub(s, rf"default command \(64-step launches, 128 GiB table\): \*\*{NUM}·10¹⁰ env-steps/s, {NUM}\*\*",
        f"default command (64-step launches, 128 GiB table): **{vd:.2f}·10¹⁰ env-steps/s, {fd:.3f}**", "README default")
s = sub(s, rf"the bench's frozen-table companion: {NUM}·10¹⁰, {NUM} µs per step, {NUM};",
        f"the bench's frozen-table companion: {fz20['value'] / 1e10:.2f}·10¹⁰, {fz20['ms_per_step'] * 1e3:.1f} µs per step, {fz20['roofline_frac']:.3f};", "README frozen")
s = sub(s, rf"the oracle \(C port of the reference\): {NUM}·10[⁰-⁹]+ on one thread, {NUM}·10[⁰-⁹]+ on \d+; \*\*the product's own CPU twin\*\* \(([^)]*)\): {NUM}·10[⁰-⁹]+ on one thread, {NUM}·10[⁰-⁹]+ on \d+",
        lambda m: (f"the oracle (C port of the reference): {sci(cb['single_thread']['value'])} on one thread, {sci(cb['value'])} on {cb['cores']}; "
                   f"**the product's own CPU twin** ({m.group(1)}): {sci(pc['single_thread']['value'])} on one thread, {sci(pc['value'])} on {pc['cores']}"),
        "README cpu baselines")
s = sub(s, rf"\| 5×5 boards, 1,048,576 envs \| {NUM}–{NUM}·10¹⁰ env-steps/s \({NUM} on 156 B/step",
        f"| 5×5 boards, 1,048,576 envs | {c5['value'] / 1e10:.2f}–{c5d['value'] / 1e10:.2f}·10¹⁰ env-steps/s ({c5['roofline_frac']:.3f} on 156 B/step", "README 5x5")
s = sub(s, rf"— {NUM} on 64-step ones\)", f"— {c5d['roofline_frac']:.3f} on 64-step ones)", "README 5x5 default")
s = sub(s, r"\| [0-9 ]+ iterations/s on the GPU", f"| {ads} iterations/s on the GPU", "README adapters")
s = sub(s, rf"\| {NUM}·10¹⁰ env-steps/s \({NUM} µs per 1 Mi-board step\) \|", f"| {1048576 / fc1 / 1e4:.2f}·10¹⁰ env-steps/s ({fc1:.1f} µs per 1 Mi-board step) |", "README four-call")
s = sub(s, rf"\| {NUM}·10⁹ env-steps/s \({NUM} µs per 1 Mi-board step in six launches\)",
        f"| {sci(1048576 / det1 * 1e6)} env-steps/s ({det1:.1f} µs per 1 Mi-board step in six launches)", "README det")
open(P("README.md"), "w", encoding="utf-8").write(s)

# ---- INTEGRATION.md -----------------------------------------------------------------------------------------------
s = open(P("INTEGRATION.md"), encoding="utf-8").read()
s = sub(s, r"\*\*[0-9 ]+ loop iterations/s\*\*", f"**{ads} loop iterations/s**", "INTEGRATION adapters")
open(P("INTEGRATION.md"), "w", encoding="utf-8").write(s)

# ---- the manifest's phrases that embed a neighbouring figure ---------------------------------------------------------
m = json.load(open(P("profiles", "quoted_figures.json"), encoding="utf-8"))
for e in m:
    q = e["quote"]
    if e["doc"] == "README.md":
        q = re.sub(rf"^{NUM} µs per step, \{{\}} of the 8 TB/s", f"{ms20:.1f} µs per step, {{}} of the 8 TB/s", q)
        q = re.sub(rf"env-steps/s, {NUM} µs per step$", f"env-steps/s, {ms20:.1f} µs per step", q)
        q = re.sub(rf"µs per step, {NUM} of the 8 TB/s$", f"µs per step, {f20:.3f} of the 8 TB/s", q)
        q = re.sub(rf"^\*\*{NUM}·10¹⁰ env-steps/s, \{{\}}\*\*$", f"**{vd:.2f}·10¹⁰ env-steps/s, {{}}**", q)
        q = re.sub(rf"^\*\*\{{\}}·10¹⁰ env-steps/s, {NUM}\*\*$", f"**{{}}·10¹⁰ env-steps/s, {fd:.3f}**", q)
    if e["doc"] == "DESIGN.md":
        q = re.sub(rf"µs per step, launch {NUM} ms$", f"µs per step, launch {l20:.3f} ms", q)
        q = re.sub(rf"env-steps/s, {NUM} µs per step$", f"env-steps/s, {ms20:.1f} µs per step", q)
        q = re.sub(rf"ms per launch = {NUM}$", f"ms per launch = {fd:.3f}", q)
        q = re.sub(rf"^= {NUM}, `fabric_frac`", f"= {fd:.3f}, `fabric_frac`", q)
    e["quote"] = q
json.dump(m, open(P("profiles", "quoted_figures.json"), "w", encoding="utf-8"), indent=1, ensure_ascii=False)
print(f"k20 {v20:.2f}e10 {ms20:.1f} us {f20:.3f} fabric {ff20:.3f} | default {vd:.2f}e10 {fd:.3f} | four-call {fc1} | det {det1} | adapters {ads}")
