"""Merge a freshly tuned tile table into the committed one, conservatively: take NEW shapes and entries that move to a halo-tile
instance (pipe 8); every other existing entry keeps its committed plan (re-timing noise must not reshuffle split-K factors, which
change fp32 rounding).   python tools/merge_tile_table.py <tuned.json> [--all]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "leftrefill_amd", "tile_table.json")
old = json.load(open(dst))
new = json.load(open(sys.argv[1]))
take_all = "--all" in sys.argv
n_new = n_halo = 0
for k, v in new.items():
    if k not in old:
        old[k] = v
        n_new += 1
    elif (take_all or (len(v) > 3 and v[3] == 8)) and list(old[k]) != list(v):
        print(k, old[k], "->", v)
        old[k] = v
        n_halo += 1
json.dump({k: list(v) for k, v in sorted(old.items())}, open(dst, "w"), indent=0)
print(f"{n_new} new shapes, {n_halo} changed entries, {len(old)} total")
