import sys, numpy as np
sys.path.insert(0, 'tests')
import oracle_lib as O
import test_oracle_physics as T
rs = np.random.RandomState(5)
hb = np.array([0.015, 0.015, 0.015]); r, hl = 0.03, 0.01
tilt = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = []
for trial in range(N):
    yaw = T._rot_axis(np.array([0.0, 0.0, 1.0]), rs.uniform(0, 2 * np.pi))
    Ra = T._rot_axis(rs.normal(size=3), rs.uniform(0, tilt))
    Rb = T._rot_axis(rs.normal(size=3), rs.uniform(0, tilt)) @ yaw
    cb = rs.uniform(-0.1, 0.1, 3)
    u = rs.normal(size=3); u /= np.linalg.norm(u)
    want = rs.uniform(2e-4, 1.8e-3)
    lo, hi = 0.0, 0.2
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        p, q = T._closest_pair(cb + u * mid, Ra, (r, hl), cb, Rb, hb, iters=150)
        if np.linalg.norm(p - q) > want: hi = mid
        else: lo = mid
    ca = cb + u * hi
    p, q = T._closest_pair(ca, Ra, (r, hl), cb, Rb, hb, iters=20000)
    gap = np.linalg.norm(p - q)
    if not (1e-4 < gap < 1.95e-3): continue
    c = O.cyl_box(ca, Ra.ravel(), r, hl, cb, Rb.ravel(), hb)
    if len(c) == 0:
        rows.append((np.nan, 'miss', 0, 0)); continue
    n = c[0, 6:9]; a = Ra[:, 2]
    kind = 'axis' if abs(n @ a) > 0.99999 else ('face' if np.abs(Rb.T @ n).max() > 0.99999 else 'other')
    rows.append((c[:, 9].min() - gap, kind, abs(n @ a), len(c)))
err = np.array([x[0] for x in rows])
print('poses', len(rows), 'beyond 1e-4:', (np.abs(err) > 1e-4).sum(), 'beyond 2e-4:', (np.abs(err) > 2e-4).sum(), 'worst', np.nanmin(err), np.nanmax(err))
for e, k, can, m in rows:
    if not (abs(e) <= 1e-4): print('%+.2e %s |n.a| %.4f contacts %d' % (e, k, can, m))
import collections
print(collections.Counter(k for _, k, _, _ in rows))
