import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
a = a[a[:, 3] > 0]
t0 = a[:, 1].min()
start, mid, end = a[:, 1] - t0, a[:, 2] - t0, a[:, 3] - t0
print("waves", len(a), "kernel span %.1f us" % (end.max() / 100))
comp = (mid - start) / 100.0
print("compute phase per wave (us): mean %.1f p50 %.1f p90 %.1f max %.1f" % (comp.mean(), np.percentile(comp, 50), np.percentile(comp, 90), comp.max()))
wg = a[:, 0] // 16
nwg = wg.max() + 1
wg_start = np.array([start[wg == g].min() for g in range(nwg)]) / 100.0
wg_mid = np.array([mid[wg == g].max() for g in range(nwg)]) / 100.0
wg_end = np.array([end[wg == g].max() for g in range(nwg)]) / 100.0
wg_meancomp = np.array([comp[wg == g].mean() for g in range(nwg)])
print("WG: start mean %.1f max %.1f; slowest-wave compute mean %.1f; mean-wave compute mean %.1f; epilogue mean %.1f; end max %.1f" % (
    wg_start.mean(), wg_start.max(), (wg_mid - wg_start).mean(), wg_meancomp.mean(), (wg_end - wg_mid).mean(), wg_end.max()))
print("WG start histogram (us):", np.histogram(wg_start, bins=8)[0].tolist(), np.histogram(wg_start, bins=8)[1].round(1).tolist())
print("WG duration (us): mean %.1f p90 %.1f max %.1f" % ((wg_end - wg_start).mean(), np.percentile(wg_end - wg_start, 90), (wg_end - wg_start).max()))
kb = (np.arange(nwg) % 64)
print("WG duration by rank block (every 8th):", [(round(float((wg_end - wg_start)[kb == k].mean()), 1)) for k in range(0, 64, 8)])
