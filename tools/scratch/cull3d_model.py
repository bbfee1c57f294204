"""How many (row, 256-column tile) pairs survive the gap culling of bitmask_rec3d_culled_kernel, for column orders: x only (now),
z strips x x, and with an additional z-gap test."""
import sys, numpy as np
sys.path.insert(0, ".")
from groomed_nms_amd import synthetic
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
p, s = synthetic.batch_3d(1000, 1, N, True)
c = O.corners_of_cuboid(p[0])                      # [N,3,8]
x0, x1 = c[:, 0, :].min(1), c[:, 0, :].max(1)
z0, z1 = c[:, 2, :].min(1), c[:, 2, :].max(1)
lx, lz = x1 - x0, z1 - z0
thr = 0.4
kappa = max(1 / (2 * thr) - 1, 0) + 1e-3
rank = np.argsort(-s[0], kind="stable")
def survivors(order, use_z):
    tot = 0
    for t in range(0, N, 256):
        cols = order[t:t + 256]
        hx0, hx1, mlx = x0[cols].min(), x1[cols].max(), lx[cols].max()
        hz0, hz1, mlz = z0[cols].min(), z1[cols].max(), lz[cols].max()
        minrank_ok = np.ones(N, bool)
        gx = np.maximum(hx0 - x1, x0 - hx1)
        skip = (gx >= 0) & (gx >= (lx + mlx) * kappa)
        if use_z:
            gz = np.maximum(hz0 - z1, z0 - hz1)
            skip |= (gz >= 0) & (gz >= (lz + mlz) * kappa)
        tot += int((~skip).sum())
    return tot / (N * (N / 256))
xc = 0.5 * (x0 + x1); zc = 0.5 * (z0 + z1)
print("x order, x test      :", round(survivors(np.argsort(xc, kind="stable"), False), 4))
print("x order, x+z test    :", round(survivors(np.argsort(xc, kind="stable"), True), 4))
for S in (2, 4, 8, 16):
    strip = np.minimum(((zc - zc.min()) / (zc.max() - zc.min() + 1e-9) * S).astype(int), S - 1)
    order = np.lexsort((xc, strip))
    print("strips", S, "x+z test:", round(survivors(order, True), 4))
