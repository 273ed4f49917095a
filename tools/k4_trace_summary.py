#!/usr/bin/env python3
"""Summarise gpurun_out/k4_trace.txt (GPIS_K4_TRACE=<block> tools/k4_bench.py ...).
Rows 0..W-1: per-wave phase stamps; rows W..2W-1: owner-path stamps of wave (row - W):
per owned block [before A-operand wait, operands arrived, update done + diagonal tile landed, solved, published]."""
import sys
lines = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/k4_trace.txt').read().split('\n')
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
detail = int(sys.argv[3]) if len(sys.argv) > 3 else -1
waves = {}; cur = None
for l in lines:
    if l.startswith('wave'):
        cur = int(l.split()[1]); waves[cur] = []
    elif l.strip():
        waves[cur].append(int(l))
for w in range(W):
    ts = waves.get(w, [])
    if len(ts) < 5:
        continue
    print('wave', w, 'events', len(ts), 'total cycles', ts[-1] - ts[0])
    print('  stage0 %d  exp %d  bgen %d' % (ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2]))
    steps = ts[4:-1] if (len(ts) - 5) % 4 == 0 else ts[4:]
    n = len(steps) // 4
    waitt = sum(steps[4 * i + 1] - steps[4 * i] for i in range(n)); own = sum(steps[4 * i + 2] - steps[4 * i + 1] for i in range(n))
    gen = sum(steps[4 * i + 3] - steps[4 * i + 2] for i in range(n)); gap = sum(steps[4 * (i + 1)] - steps[4 * i + 3] for i in range(n - 1))
    print('  steps %d: wait %d  owner(update+solve) %d  general updates %d  inter-step %d' % (n, waitt, own, gen, gap))
    if w == detail:
        for i in range(n):
            print('    c=%d wait %d owner %d general %d' % (i, steps[4 * i + 1] - steps[4 * i], steps[4 * i + 2] - steps[4 * i + 1], steps[4 * i + 3] - steps[4 * i + 2]))
    ev = waves.get(W + w, [])
    if w == 0 and len(ev) >= 3:
        print('  block 0: solve %d publish %d' % (ev[1] - ev[0], ev[2] - ev[1])); ev = ev[3:]
    for i in range(len(ev) // 5):
        e = ev[5 * i: 5 * i + 5]
        print('  owned block %d (t=%d): operand wait %d  update %d  solve %d  publish %d' % (i, e[0], e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]))
