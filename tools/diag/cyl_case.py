import sys, numpy as np
sys.path.insert(0, 'tests')
import oracle_lib as O
import test_oracle_physics as T
rs = np.random.RandomState(5)
hb = np.array([0.015, 0.015, 0.015]); r, hl = 0.03, 0.01
tilt = 0.05
for trial in range(300):
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
    if len(c) and abs(c[:, 9].min() - gap) > 1e-4:
        print('gap', gap, 'reported', c[:, 9], 'n', c[0, 6:9])
        # local frames
        pl = Ra.T @ (p - ca); ql = Rb.T @ (q - cb)
        print(' true pair: cyl local', pl, 'rho', np.hypot(pl[0], pl[1]), ' box local', ql / hb)
        for cc_ in c:
            print(' contact pa cyl-local', Ra.T @ (cc_[0:3] - ca), 'pb box-local', (Rb.T @ (cc_[3:6] - cb)) / hb)
        d = (p - q) / gap
        print(' true dir . a', d @ Ra[:, 2], ' true dir in box', Rb.T @ d)
