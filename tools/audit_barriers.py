#!/usr/bin/env python
"""Audit of a gfx950 assembly listing (hipcc -S): every s_barrier whose nearest preceding LDS write (ds_write* / buffer_load ... lds
is not checked: that one is covered by vmcnt) in the linear listing is NOT followed by an s_waitcnt with lgkmcnt(0) before the
barrier.  Round 6: behind inline-asm s_waitcnt statements hipcc dropped the LDS wait of __syncthreads() and other waves read a
tile whose last ds_write was still in flight.  Linear scan (labels and branches are not followed): a hit is a place to read, not
a proof.  usage: audit_barriers.py file.s [file.s ...]"""
import re
import sys

for path in sys.argv[1:]:
    kernel = "?"
    lines = open(path).read().split("\n")
    last_write = None          # line index of the last ds_write not yet covered by an lgkmcnt(0) wait
    for k, l in enumerate(lines):
        t = l.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel, last_write = m.group(1), None
        if t.startswith("ds_write") or t.startswith("ds_store"):
            last_write = k
        elif t.startswith("s_waitcnt") and ("lgkmcnt(0)" in t):
            last_write = None
        elif t.startswith("s_barrier") and last_write is not None:
            print(f"{path}:{k + 1}: s_barrier {k - last_write} lines after an uncovered ds_write (line {last_write + 1}) in {kernel[:90]}")
            last_write = None
