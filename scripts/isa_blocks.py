#!/usr/bin/env python3
"""Instruction mix of a kernel's ISA listing, block by block: for every straight-line stretch between labels and branches
of the listing range [first, last] print the VALU / SALU / LDS / vector-memory counts and the branch that ends it.
usage: isa_blocks.py listing.s first_line last_line   (listing = hipcc -S --cuda-device-only output)"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
lo, hi = int(sys.argv[2]), int(sys.argv[3])
cnt = dict(v=0, s=0, l=0, m=0, w=0)
start = lo


def flush(i, why):
    global cnt, start
    if sum(cnt.values()):
        print("%5d-%5d  VALU %3d  SALU %3d  LDS %2d  VMEM %2d  wait %d   %s" % (start, i, cnt["v"], cnt["s"], cnt["l"], cnt["m"], cnt["w"], why))
    elif why:
        print("%5d        %s" % (i, why))
    cnt = dict(v=0, s=0, l=0, m=0, w=0)
    start = i + 1


for i in range(lo, hi + 1):
    l = lines[i - 1].strip()
    if not l or l.startswith(";"):
        continue
    if re.match(r"^[.\w]+:", l):
        flush(i - 1, "")
        print("%5d  %s" % (i, l.split()[0]))
        start = i
        continue
    op = l.split()[0]
    if op.startswith("s_cbranch") or op == "s_branch":
        cnt["s"] += 1
        flush(i, l.split(";")[0].strip())
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
        cnt["w"] += 1
    elif op.startswith("s_"):
        cnt["s"] += 1
    elif op.startswith("v_"):
        cnt["v"] += 1
    elif op.startswith("ds_"):
        cnt["l"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")):
        cnt["m"] += 1
flush(hi, "")
