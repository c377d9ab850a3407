#!/usr/bin/env python3
"""Condense a tools/multi_modes.sh log: one line per (library, mode, prep, K) with the us per batch of every repetition."""
import re
import sys
h, res, bad = None, {}, 0
for ln in open(sys.argv[1]):
    if ln.startswith('==') or ln.startswith('##'):
        h = ln.strip()
        res.setdefault(h, [])
    m = re.search(r'launches: ([0-9.]+) us per batch.*frac ([0-9.]+)', ln)
    if m:
        res[h].append((float(m.group(1)), float(m.group(2))))
    m = re.search(r'isolated launches.*median ([0-9.]+)  p75 ([0-9.]+)  max ([0-9.]+)', ln)
    if m:
        res[h].append((float(m.group(1)), -1.0))
    if 'mismatch' in ln and ' 0 mismatches' not in ln:
        bad += 1
        print("BAD", h, ln.strip())
    if 'FAILED' in ln or 'error' in ln.lower():
        print(ln.strip())
for k, v in res.items():
    if v:
        print(f"{k:56s}", " ".join(f"{a:.3f}" for a, _ in v), " best frac", max(f for _, f in v))
    else:
        print(k)
print("mismatching checks:", bad)
