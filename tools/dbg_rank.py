import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shermbot-navigation_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nuslam_hip as hip
from nuslam_hip import synth
Q, R = synth.Q_DEFAULT, synth.R_DEFAULT
n, m, T = 40, 16, 12
tr = synth.make_trace(n, T, m, straight_every=3)
a = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R); b = hip.EKF(np.zeros(3), np.zeros(2 * n), Q, R)
a.as_batch().set_tick_mode(1); a.as_batch().set_pass_variant(hip.PASS_RANK)
b.as_batch().set_tick_mode(1); b.as_batch().set_pass_variant(hip.PASS_EXACT)
seen_ids = set()
for t in range(T):
    a.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t]); b.tick(tr.tw[t], tr.mx[t], tr.my[t], known_ids=tr.ids[t])
    new = [int(i) for i in tr.ids[t] if int(i) not in seen_ids]
    seen_ids.update(int(i) for i in tr.ids[t])
    sa, sb, Pa, Pb = a.state, b.state, a.cov, b.cov
    ds = np.abs(sa - sb); dP = np.abs(Pa - Pb)
    i = int(ds.argmax()); ij = np.unravel_index(dP.argmax(), dP.shape)
    d = np.diag(Pb)
    print("tick %2d new %s seen %d | state max abs %.2e at %d (val %.3e) | cov max abs %.2e at %s (val %.3e) | max finite diag %.3e n_intmax %d"
          % (t, new, a.seen, ds.max(), i, sb[i], dP.max(), ij, Pb[ij], d[d < 1e9].max(), (d > 1e9).sum()))
