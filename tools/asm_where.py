#!/usr/bin/env python3
"""Where in the source do instructions matching a pattern sit?  usage: asm_where.py <file.s (built with -g1)> <regex> [function substring]
Prints file:line (from the nearest preceding .loc) with a count, most frequent first."""
import re, sys, collections
path, pat = sys.argv[1], re.compile(sys.argv[2])
fn_filter = sys.argv[3] if len(sys.argv) > 3 else None
files = {}
cur_fn, loc = None, None
counts = collections.Counter()
for l in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
        continue
    m = re.match(r"^(_Z[\w]+):", l)
    if m:
        cur_fn = m.group(1)
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = (int(m.group(1)), int(m.group(2)))
        continue
    if pat.search(l) and not l.strip().startswith((";", ".")):
        if fn_filter and (cur_fn is None or fn_filter not in cur_fn):
            continue
        f = files.get(loc[0], "?") if loc else "?"
        counts[(cur_fn[-40:] if cur_fn else "?", f.split("/")[-1], loc[1] if loc else 0)] += 1
for (fn, f, line), n in counts.most_common(60):
    print("%5d  %-42s %s:%d" % (n, fn, f, line))
