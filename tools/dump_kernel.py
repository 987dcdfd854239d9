#!/usr/bin/env python3
"""Dump the gfx950 disassembly of the kernels of a library / object whose (mangled) name contains every given substring.
usage: tools/dump_kernel.py lib.so rollout_wt_kernel ELi0ELi10ELi200 [...] > out.s      (one '=== symbol' header per kernel)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_isa_hygiene as t  # noqa: E402

path, keys = sys.argv[1], sys.argv[2:]
for img in t._code_objects(path):
    for sym, ins in t._kernels(img, keys[0]).items():
        if all(k in sym for k in keys[1:]):
            print("=== %s (%d lines)" % (sym, len(ins)))
            print("\n".join(ins))
