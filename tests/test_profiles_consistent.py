"""Prose that equals the evidence (`profiles/README.md` quoted 461.10 us where the tracked CSV said
468.5, DESIGN.md 19.43 ms where the tracked trace said 20.25 - the last session overwrote the files, the prose kept an earlier one).

* The round's figures are RENDERED from the tracked files by tools/profile_docs.py into a marked block of profiles/README.md and
  DESIGN.md; this test re-renders the block and fails when either document holds anything else.
* Every duration the round's prose quotes outside that block - the `r06/` section of profiles/README.md, and the parts of DESIGN.md
  between `<!-- round6:begin -->` / `<!-- round6:end -->` - must be a number the tracked evidence holds (a figure of the block, or a
  duration printed in one of profiles/r06/*.txt), within 0.5 %.
CPU-only: it reads files.
"""
import glob
import importlib.util
import os
import re

import pytest

from conftest import ROOT

RND = "r06"
DUR = re.compile(r"(?<![\w.])(\d+(?:[ ,]\d{3})*(?:\.\d+)?)\s*(µs|us|ms)\b")


def _docs():
    spec = importlib.util.spec_from_file_location("profile_docs", os.path.join(ROOT, "tools", "profile_docs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _num(s):
    return float(s.replace(" ", "").replace(",", ""))


def _evidence_us():
    """Every duration the tracked evidence of the round holds, in microseconds."""
    pd = _docs()
    f = pd.figures(RND)
    pool = [f["headline"]["avg_us"], f["headline"]["min_us"], f["headline"]["max_us"], f["fixup_us"]]
    for v in list(f["workloads"].values()) + list(f["others"].values()):
        pool += [v["average_ms"] * 1e3, v["min_ms"] * 1e3]
        if v.get("second_launch_us"):
            pool.append(v["second_launch_us"])
    for name in ("bench_driver_cmd.json", "bench_unprofiled.json"):
        rec = f.get(name)
        if rec:
            pool += [rec["ms_per_step"] * 1e3, rec["roofline"]["kernel_ms"] * 1e3]
            pool += [v * 1e3 for k, v in rec.get("extras", {}).items() if k.endswith("_ms")]
            pool += [v for k, v in rec.get("extras", {}).items() if k.endswith("_us_per_call")]
    def unit_of(key, inherited):
        if re.search(r"(^|_)ms($|_)", key):
            return 1e3
        if re.search(r"(^|_)us($|_)", key):
            return 1.0
        return inherited

    def walk(node, unit=None):                                # the bench records of the session: every leaf under a key that names a duration
        if isinstance(node, dict):
            for k, v in node.items():
                u = unit_of(k, unit)
                if isinstance(v, (dict, list)):
                    walk(v, u)
                elif isinstance(v, (int, float)) and not isinstance(v, bool) and u:
                    pool.append(v * u)
        elif isinstance(node, list):
            for v in node:
                walk(v, unit)
    import json
    for name in ("bench_driver_detail.json", "bench_unprofiled.json"):
        path = os.path.join(ROOT, "profiles", RND, name)
        if os.path.exists(path):
            walk(json.loads(open(path).read()))
    for path in glob.glob(os.path.join(ROOT, "profiles", RND, "**", "*.txt"), recursive=True):
        for m in DUR.finditer(open(path, errors="replace").read()):
            pool.append(_num(m.group(1)) * (1e3 if m.group(2) == "ms" else 1.0))
    return sorted(pool)


def _check(text, where, pool):
    import bisect
    bad = []
    for m in DUR.finditer(text):
        us = _num(m.group(1)) * (1e3 if m.group(2) == "ms" else 1.0)
        if us == 0:
            continue
        digits = len(m.group(1).split(".")[1]) if "." in m.group(1) else 0
        half_ulp = 0.5 * 10 ** -digits * (1e3 if m.group(2) == "ms" else 1.0)      # (a figure written with n decimals stands for +- half a unit of the last one)
        tol = max(0.005 * us, half_ulp)
        i = bisect.bisect_left(pool, us - tol)
        if not (i < len(pool) and pool[i] <= us + tol):
            bad.append(f"{where}: '{m.group(0)}' is in no tracked file of profiles/{RND}/ (context: ...{text[max(0, m.start() - 60):m.end() + 20]!r}...)")
    return bad


def test_generated_blocks_equal_the_tracked_files():
    pd = _docs()
    block = pd.block(RND)
    for doc in pd.DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        m = re.search(rf"<!-- profiles:{RND} begin -->.*?<!-- profiles:{RND} end -->", text, re.S)
        assert m, f"{doc} has no profiles:{RND} block"
        assert m.group(0) == block, f"{doc}: the profiles:{RND} block is not what the tracked files render to - run `python tools/profile_docs.py {RND} --write`"


def test_traffic_json_is_this_rounds_reduction():
    import csv
    import json
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    with open(os.path.join(ROOT, "profiles", RND, "bench_kernel_stats.csv")) as fh:
        row = next(r for r in csv.DictReader(fh) if t["kernel"] in r["Name"])
    assert abs(float(row["AverageNs"]) - t["kernel_trace"]["average_ns"]) < 1e-3 and int(row["Calls"]) == t["kernel_trace"]["calls"]
    assert f"profiles/{RND}/" in t["method"]


def test_every_duration_the_rounds_prose_quotes_is_in_the_evidence():
    pool = _evidence_us()
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    m = re.search(rf"`{RND}/`.*?(?=\n`r\d\d/`)", readme, re.S)
    assert m, f"profiles/README.md has no `{RND}/` section"
    section = re.sub(rf"<!-- profiles:{RND} begin -->.*?<!-- profiles:{RND} end -->", "", m.group(0), flags=re.S)
    bad = _check(section, "profiles/README.md", pool)
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    parts = re.findall(r"<!-- round6:begin -->(.*?)<!-- round6:end -->", design, re.S)
    assert parts, "DESIGN.md marks its round-6 prose with <!-- round6:begin --> ... <!-- round6:end -->"
    for part in parts:
        bad += _check(part, "DESIGN.md", pool)
    assert not bad, "\n".join(bad)


def test_a_stale_figure_would_be_caught():
    """The check itself, on last round's mismatch: 461.10 us quoted where the tracked CSV said 468.52 fails; the tracked figure, written
    with fewer decimals, passes."""
    pool = [468.516857]
    assert not _check("**468.5 µs** (210 launches)", "x", pool) and not _check("0.4685 ms", "x", pool)
    assert _check("**461.10 µs** (455.4-475.8)", "x", pool)
