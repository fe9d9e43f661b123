#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S): the blocks of more
than `min` instructions with their opcode histogram -- how many instructions a loop body costs.   python tools/loop_insts.py file.s name [min]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
name, mn = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l and ': ' in l and '@' in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur = [], None
for l in lines[start:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = [l.split(':')[0], 0, {}]
        blocks.append(cur)
    elif cur is not None and l.strip() and not l.strip().startswith((';', '.')):
        cur[1] += 1
        op = l.split()[0]
        cur[2][op] = cur[2].get(op, 0) + 1
for b in blocks:
    if b[1] >= mn:
        print(b[0], b[1])
        print('  ', sorted(b[2].items(), key=lambda x: -x[1]))
