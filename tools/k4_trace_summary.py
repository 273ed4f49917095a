#!/usr/bin/env python3
"""Per-class cycle breakdown of K4 from gpurun_out/k4_trace.txt (instrumented build: EXTRA=-DGPIS_INSTRUMENT,
gpismap_amd/csrc/ongpis_test_instr.inc).  One line per kernel class (wavefronts per workgroup, exp table or not): where the
cycles of the sampled workgroups go, as a share of the workgroup's wave-cycles (sum over its wavefronts of start -> end):

  stage   vectors + queries into LDS           exp     exp table                  first   first chunk(s) generated + barrier
  gen     B-tile generation in the main loop   mma     X loads + matrix instrs    bar     barrier at the end of every chunk
  ss      sums of squares per row group        tail    cross-wave reduce + store

usage: tools/k4_trace_summary.py [trace.txt] [min_tiles]"""
import collections
import sys


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/k4_trace.txt"
    min_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    classes = collections.OrderedDict()
    cur = None
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "launch":
            d = dict(zip(t[1::2], map(int, t[2::2])))
            cur = None
            if d["tiles"] >= min_tiles:
                cur = classes.setdefault((d["W"], d["table"]), dict(launches=0, tiles=0, maxLd=0, wg={}, seq=0))
                cur["launches"] += 1; cur["tiles"] += d["tiles"]; cur["maxLd"] = max(cur["maxLd"], d["maxLd"]); cur["seq"] += 1
        elif t[0] == "wg" and cur is not None:
            wg, wave = int(t[1]), int(t[3])
            st = list(map(int, t[5:13])); laps = list(map(int, t[14:19]))
            cur["wg"].setdefault((cur["seq"], wg), {})[wave] = (st, laps)
    print("%-18s %8s %9s %7s %10s | %6s %6s %6s | %6s %6s %6s %6s | %6s | %7s %s" %
          ("class", "launches", "tiles", "maxK", "WG cycles", "stage", "exp", "first", "gen", "mma", "bar", "ss", "tail", "chunks", "slowest-wave mma share"))
    for (W, table), c in classes.items():
        tot = collections.Counter(); nwg = 0; wgc = 0.0; chunks = 0.0; crit = 0.0
        for waves in c["wg"].values():
            if not waves:
                continue
            nwg += 1
            end = max(st[5] for st, _ in waves.values())
            wgc += end
            best = 0.0
            for st, laps in waves.values():
                tot["all"] += st[5]
                tot["stage"] += st[1]; tot["exp"] += st[2] - st[1]; tot["first"] += st[3] - st[2]
                tot["gen"] += laps[0]; tot["mma"] += laps[1]; tot["bar"] += laps[2]; tot["ss"] += laps[3]
                tot["tail"] += st[5] - st[4]
                chunks += laps[4]
                best = max(best, laps[1] / max(1, st[5]))
            crit += best
        if not nwg:
            continue
        a = float(tot["all"]) or 1.0
        pct = lambda k: 100.0 * tot[k] / a
        print("W=%d %-13s %8d %9d %7d %10.0f | %5.1f%% %5.1f%% %5.1f%% | %5.1f%% %5.1f%% %5.1f%% %5.1f%% | %5.1f%% | %7.1f %5.1f%%   (%d workgroups sampled)" %
              (W, "table" if table else "no table", c["launches"], c["tiles"], c["maxLd"] - 1, wgc / nwg, pct("stage"), pct("exp"), pct("first"),
               pct("gen"), pct("mma"), pct("bar"), pct("ss"), pct("tail"), chunks / max(1, sum(len(w) for w in c["wg"].values())), 100.0 * crit / nwg, nwg))


if __name__ == "__main__":
    main()
