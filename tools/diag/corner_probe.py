import sys, numpy as np
sys.path.insert(0,'tests')
import oracle_lib as O
I = np.eye(3).ravel()
cc = np.array([-0.495, 0.0979, 0.17]); r, hl = 0.03, 0.01
hb = np.array([0.0125, 0.005, 0.04])
for yoff in (0.0, 0.0157, 0.0257):    # finger-box centre y relative to the puck centre
    print('finger box centre y offset', yoff)
    for gap in np.arange(0.050, 0.028, -0.002):
        cb = np.array([cc[0] + gap, cc[1] + yoff, 0.177 + 0.03])
        c = O.cyl_box(cc, I, r, hl, cb, I, hb)
        # analytic: nearest point of the box (in xy) to the cylinder axis
        dx = max(abs(cb[0] - cc[0]) - hb[0], 0); dy = max(abs(cb[1] - cc[1]) - hb[1], 0)
        true = np.hypot(dx, dy) - r
        print('  gap %.3f true distance %+.5f  contacts %d  reported %s  n %s' % (gap, true, len(c), np.round(c[:, 9], 5) if len(c) else '-', np.round(c[0, 6:9], 3) if len(c) else '-'))
