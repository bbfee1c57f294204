import sys, numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
a = a[a[:, 2] > 0]
start, end = a[:, 1], a[:, 2]
dur = end - start
span = end.max()
print("waves", len(a), "span", span, "ticks; duration ticks: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
# resident waves over time
edges = np.linspace(0, span, 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    print("t=%5.1f%%  resident %5d  started %5d" % (100 * mid / span, int(((start <= mid) & (end > mid)).sum()), int(((start >= lo) & (start < hi)).sum())))
# duration by kbg group (first image): slot -> tile -> kbg
slot = a[:, 0]
per_img = len(a) // 8 if len(a) >= 8 else len(a)
img = slot // per_img
tile = slot % per_img
nchunk = 64
kbg, chunk = tile // nchunk, tile % nchunk
print("mean duration by image:", [int(dur[img == i].mean()) for i in range(img.max() + 1)])
print("mean start by image   :", [int(start[img == i].mean()) for i in range(img.max() + 1)])
print("mean duration by kbg (of 64):", [int(dur[kbg == k].mean()) for k in range(0, kbg.max() + 1, 4)])
print("mean duration by chunk (of 64):", [int(dur[chunk == c].mean()) for c in range(0, 64, 4)])
print("max duration by chunk :", [int(dur[chunk == c].max()) for c in range(0, 64, 4)])
late = start > 0.6 * span
print("late waves: n", int(late.sum()), "mean dur", int(dur[late].mean()), " early mean dur", int(dur[~late].mean()))
