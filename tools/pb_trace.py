"""EXPERIMENT (round 4): phase stamps of the bf16 forward (DIMN_PREDICT_TRACE=file): per-workgroup durations of each phase, in us."""
import sys
import numpy as np

def main(path):
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 8).astype(np.float64) / 100.0      # 100 MHz -> us
    ok = t[:, 6] > 0
    t = t[ok]
    t0 = t[:, 0].min()
    names = ["loop", "act", "l2a mfma", "l2a epi", "l2b mfma", "l2b epi"]
    print("%d workgroups, kernel span %.1f us" % (len(t), t[:, 6].max() - t0))
    for j, nm in enumerate(names):
        d = t[:, j + 1] - t[:, j]
        print("  %-9s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f us" % (nm, d.mean(), *np.percentile(d, [10, 50, 90])))
    tot = t[:, 6] - t[:, 0]
    print("  %-9s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f us" % ("total", tot.mean(), *np.percentile(tot, [10, 50, 90])))
    # start times: dispatch waves
    st = np.sort(t[:, 0] - t0)
    print("  starts: first 512 within %.1f us; median start gap afterwards %.3f us" % (st[min(511, len(st) - 1)], np.median(np.diff(st[512:])) if len(st) > 600 else -1))

if __name__ == "__main__":
    main(sys.argv[1])
