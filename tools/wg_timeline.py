"""Developer aid: summarise a TOPOLOW_WG_STAMPS dump (tuning build): dispatch ramp, per-workgroup
duration, tail.  Times in microseconds (100 MHz stamps)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.uint64).astype(np.int64)
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
d = en - st
q = lambda x: np.percentile(x, [0, 10, 50, 90, 100]).round(2).tolist()
print("workgroups", len(a), "span", en.max().round(2))
print("start  pct[0,10,50,90,100]", q(st))
print("end    pct[0,10,50,90,100]", q(en))
print("dur    pct[0,10,50,90,100]", q(d))
order = np.argsort(st)
print("start by blockIdx deciles", [round(float(st[i]), 2) for i in range(0, len(a), max(1, len(a) // 10))])
