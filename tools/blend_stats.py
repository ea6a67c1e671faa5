"""Work statistics of the blend kernels on a synthetic scene (CPU, numpy; uses the oracle's forward output).

For every 8x8-pixel wave: how many list entries it walks, how many survive the circle cull (rcut2), how many would
survive an exact ellipse-vs-rectangle cull, how many have at least one pixel with alpha >= 1/255 ("any"), and the
number of (splat, pixel) pairs that really blend.  Test/analysis tool only (imports oracle/)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import torch
from helpers import scene_inputs, oracle_forward
from oracle.oracle import Oracle

P, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 800, 800
scale_mult = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
act, rs, cam = scene_inputs(P, W, H, seed=0, scale_mult=scale_mult)
o = Oracle('f32')
ref = oracle_forward(o, act, rs)
geom, binning, img = ref['geom'], ref['binning'], ref['img']
xy = geom['means2D'].astype(np.float64)
con = geom['conic_opacity'].astype(np.float64)
ranges = binning['ranges'].astype(np.int64)
plist = binning['point_list'].astype(np.int64)
ncontrib = img['n_contrib'].astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
R = ref['num_rendered']
a, b, c, op = con[:, 0], con[:, 1], con[:, 2], con[:, 3]
# covariance eigenvalue from the conic: cov = inv(conic)
det = a * c - b * b
cova, covc = c / det, a / det
mid = 0.5 * (cova + covc)
lam = mid + np.sqrt(np.maximum(0.1, mid * mid - 1.0 / det))
o255 = 255 * op
rcut2 = np.where(o255 > 1, 2.02 * np.log(np.maximum(o255, 1.0000001)) * lam + 0.25, -1)
thr = 2 * np.log(np.maximum(o255, 1e-30))  # d^T Q d <= thr

tot = dict(entries=0, walked_fwd=0, walked_bwd=0, circle_fwd=0, circle_bwd=0, ellipse_bwd=0, any_bwd=0, pairs_bwd=0, pairs_alpha=0)
per_tile_cost = []
for t in range(gx * gy):
    s, e = ranges[t]
    if e <= s:
        continue
    ids = plist[s:e]
    n = e - s
    tx, ty = t % gx, t // gx
    px = tx * 16 + np.arange(16)
    py = ty * 16 + np.arange(16)
    PX, PY = np.meshgrid(px, py)
    inside = (PX < W) & (PY < H)
    dx = xy[ids, 0][:, None, None] - PX[None]
    dy = xy[ids, 1][:, None, None] - PY[None]
    power = -0.5 * (a[ids][:, None, None] * dx * dx + c[ids][:, None, None] * dy * dy) - b[ids][:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[ids][:, None, None] * np.exp(np.minimum(power, 0)))
    hit = (power <= 0) & (alpha >= 1 / 255) & inside[None]
    nc = ncontrib[np.minimum(PY, H - 1), np.minimum(PX, W - 1)] * inside
    k = np.arange(n)[:, None, None]
    live = k < nc[None]
    tot['entries'] += n
    tot['pairs_alpha'] += int(hit.sum())
    tile_cost = 0
    for sub in range(4):
        x0, y0 = tx * 16 + (sub % 2) * 8, ty * 16 + (sub // 2) * 8
        sl = (slice(None), slice((sub // 2) * 8, (sub // 2) * 8 + 8), slice((sub % 2) * 8, (sub % 2) * 8 + 8))
        maxk = nc[sl[1:]].max()
        if maxk == 0:
            continue
        ddx = np.maximum(np.maximum(x0 - xy[ids, 0], xy[ids, 0] - (x0 + 7)), 0)
        ddy = np.maximum(np.maximum(y0 - xy[ids, 1], xy[ids, 1] - (y0 + 7)), 0)
        circ = ddx * ddx + ddy * ddy <= rcut2[ids]
        # exact: min over the rectangle of the quadratic form (brute force over the 64 pixels is a lower bound proxy)
        q = -2 * power[sl]
        ell = (q.reshape(n, -1).min(1) <= thr[ids]) & (o255[ids] > 1)
        walked = np.arange(n) < maxk
        h = hit[sl] & live[sl]
        anyh = h.reshape(n, -1).any(1)
        tot['walked_bwd'] += int(walked.sum())
        tot['circle_bwd'] += int((circ & walked).sum())
        tot['ellipse_bwd'] += int((ell & walked).sum())
        tot['any_bwd'] += int(anyh.sum())
        tot['pairs_bwd'] += int(h.sum())
        tile_cost += int((circ & walked).sum())
        # 4x4 quads of the wave: list length per quad if every quad walked its own compacted list
        qany = [h[:, (qd // 2) * 4:(qd // 2) * 4 + 4, (qd % 2) * 4:(qd % 2) * 4 + 4].reshape(n, -1).any(1).sum() for qd in range(4)]
        tot['quad_max'] = tot.get('quad_max', 0) + int(max(qany))
        tot['quad_sum'] = tot.get('quad_sum', 0) + int(sum(qany))
    for strip in range(2):
        sl = (slice(None), slice(strip * 8, strip * 8 + 8), slice(None))
        maxk = nc[sl[1:]].max()
        if maxk == 0:
            continue
        h = hit[sl] & live[sl]
        tot['any_strip'] = tot.get('any_strip', 0) + int(h.reshape(n, -1).any(1).sum())
        lanes = (h[:, :, :8] | h[:, :, 8:]).reshape(n, -1).sum(1)
        tot['lanes_strip'] = tot.get('lanes_strip', 0) + int(lanes.sum())
    h = hit & live
    tot['any_tile'] = tot.get('any_tile', 0) + int(h.reshape(n, -1).any(1).sum())
    per_tile_cost.append(tile_cost)
print('P', P, 'R', R, tot)
pc = np.array(per_tile_cost)
print('per-tile backward visits: mean %.0f max %d p99 %.0f sum %d' % (pc.mean(), pc.max(), np.percentile(pc, 99), pc.sum()))
print('pairs per any-visit: %.1f of 64' % (tot['pairs_bwd'] / max(tot['any_bwd'], 1)))
