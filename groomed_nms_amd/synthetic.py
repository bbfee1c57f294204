"""Seeded synthetic inputs for the parity tests and bench.py (SURVEY.md 8-d generators).

uniform-2D:   centres U([0,1760]x[0,512]), w,h ~ U(16,136)   (canvas = crop_size, scripts/config/groumd_nms.py:91)
clustered-2D: K = N/per objects, each replicated `per` times with centre jitter N(0, 0.1*size) and
              log-size jitter N(0, 0.1) -- KITTI-like: many proposals per object, groups near the cap
3D:           x~U(-30,30), y~U(0.5,2.5), z~U(5,60), l~U(3,5), w~U(1.4,2), h~U(1.3,2), ry~U(-pi,pi)
scores:       U(0,1) fp32, distinct by construction (the reference's order among ties is unspecified)
"""
import numpy as np


def uniform_boxes_2d(rng, n):
    c = np.stack([rng.uniform(0, 1760, n), rng.uniform(0, 512, n)], 1)
    wh = rng.uniform(16, 136, size=(n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)


def clustered_boxes_2d(rng, n, per=64):
    k = max(1, n // per)
    base = uniform_boxes_2d(rng, k).astype(np.float64)
    bc = (base[:, :2] + base[:, 2:]) / 2
    bs = base[:, 2:] - base[:, :2]
    which = np.arange(n) % k
    c = bc[which] + rng.normal(0, 0.1, size=(n, 2)) * bs[which]
    s = bs[which] * np.exp(rng.normal(0, 0.1, size=(n, 2)))
    out = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    return out[rng.permutation(n)]


def boxes_3d(rng, n, clustered=False, per=64):
    def draw(m):
        return np.stack([rng.uniform(-30, 30, m), rng.uniform(0.5, 2.5, m), rng.uniform(5, 60, m),
                         rng.uniform(1.4, 2.0, m), rng.uniform(1.3, 2.0, m), rng.uniform(3, 5, m),
                         rng.uniform(-np.pi, np.pi, m)], 1)   # x y z w h l ry
    if not clustered:
        return draw(n).astype(np.float32)
    k = max(1, n // per)
    base = draw(k)
    which = np.arange(n) % k
    p = base[which].copy()
    p[:, :3] += rng.normal(0, 0.15, size=(n, 3))
    p[:, 3:6] *= np.exp(rng.normal(0, 0.05, size=(n, 3)))
    p[:, 6] += rng.normal(0, 0.05, size=n)
    return p[rng.permutation(n)].astype(np.float32)


def tie_free_scores(rng, n, lo=0.0, hi=1.0):
    s = rng.uniform(lo, hi, size=n).astype(np.float32)
    # argsort-rank perturbation: nudge duplicates apart by ulps until all distinct
    for _ in range(64):
        u, idx, cnt = np.unique(s, return_index=True, return_counts=True)
        if len(u) == n:
            return s
        dup = np.ones(n, bool)
        dup[idx] = False
        s[dup] = np.nextafter(s[dup], np.float32(hi), dtype=np.float32)
    raise RuntimeError("could not make scores distinct")


def batch_2d(seed, B, N, kind="uniform", per=64):
    """(boxes [B,N,4], scores [B,N]) fp32."""
    rng = np.random.default_rng(seed)
    gen = uniform_boxes_2d if kind == "uniform" else (lambda r, n: clustered_boxes_2d(r, n, per))
    boxes = np.stack([gen(rng, N) for _ in range(B)])
    scores = np.stack([tie_free_scores(rng, N) for _ in range(B)])
    return boxes, scores


def batch_3d(seed, B, N, clustered=True, per=64):
    rng = np.random.default_rng(seed)
    params = np.stack([boxes_3d(rng, N, clustered, per) for _ in range(B)])
    scores = np.stack([tie_free_scores(rng, N) for _ in range(B)])
    return params, scores
