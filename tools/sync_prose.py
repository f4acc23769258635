#!/usr/bin/env python3
"""After a measurement session has been copied over the tracked evidence of profiles/<round>/: bring the durations the round's prose
quotes from the *.txt evidence along.  For every tracked text file whose working-tree version differs from HEAD's, the durations
("x.xxx ms") of the two versions are paired line by line; a duration of the old version that the prose quotes verbatim is replaced by
its successor - only where the old figure maps to ONE new figure over all files.  What it cannot pair (figures out of the JSON
records, rounded figures) tests/test_profiles_consistent.py still reports; those are edited by hand.

    python tools/sync_prose.py r06            # edits DESIGN.md and profiles/README.md in place, prints what it did
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DUR = re.compile(r"(?<![\w.])(\d+\.\d+) ms\b")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
    mapping, clash = {}, set()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "*.txt"))):
        rel = os.path.relpath(path, ROOT)
        try:
            old = subprocess.run(["git", "show", f"HEAD:{rel}"], cwd=ROOT, capture_output=True, text=True, check=True).stdout
        except subprocess.CalledProcessError:
            continue
        new = open(path).read()
        if old == new:
            continue
        lo, ln = old.splitlines(), new.splitlines()
        if len(lo) != len(ln):
            print(f"  (skipped {rel}: {len(lo)} lines became {len(ln)})")
            continue
        for a, b in zip(lo, ln):
            da, db = DUR.findall(a), DUR.findall(b)
            if len(da) != len(db) or DUR.sub("#", a) != DUR.sub("#", b):
                continue
            for x, y in zip(da, db):
                if x == y:
                    continue
                if x in mapping and mapping[x] != y:
                    clash.add(x)
                mapping[x] = y
    for x in clash:
        mapping.pop(x, None)
    for doc in ("DESIGN.md", "profiles/README.md"):
        p = os.path.join(ROOT, doc)
        text = open(p).read()
        n = 0

        def repl(m):
            nonlocal n
            v = m.group(1)
            if v in mapping:
                n += 1
                return m.group(0).replace(v, mapping[v])
            return m.group(0)
        # every three-decimal figure INSIDE the round's marked prose (<!-- round6:begin --> ... <!-- round6:end -->;
        # profiles/README.md: the round's section) - "a / b / c ms" and "a against b ms" runs carry the unit once
        num = re.compile(r"(?<![\w.])(\d+\.\d{3})(?![\d])")
        if doc == "DESIGN.md":
            new_text = re.sub(r"<!-- round6:begin -->.*?<!-- round6:end -->", lambda m: num.sub(repl, m.group(0)), text, flags=re.S)
        else:
            new_text = re.sub(rf"`{rnd}/`.*?(?=\n`r\d\d/`)", lambda m: num.sub(repl, m.group(0)), text, count=1, flags=re.S)
        if new_text != text:
            open(p, "w").write(new_text)
        print(f"{doc}: {n} figures moved")
    if clash:
        print("  ambiguous (left alone):", sorted(clash))


if __name__ == "__main__":
    main()
