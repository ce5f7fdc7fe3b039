import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import caltech_bifrost_dsp_amd
from caltech_bifrost_dsp_amd import ring as R
impl = sys.argv[1]; N = 20000
R.IMPLEMENTATION = impl
r = R.Ring("t", space="system"); r.resize(64, 64 * (N + 8))
gen = r.read(guarantee=True)
t0 = time.perf_counter()
w = r.begin_writing(); oseq = w.begin_sequence(time_tag=0, header="{}")
for k in range(N):
    sp = oseq.reserve(64); sp.close()
oseq.end(); w.__exit__(None, None, None)
t1 = time.perf_counter()
n = 0
for iseq in gen:
    for ispan in iseq.read(64):
        n += 1
t2 = time.perf_counter()
print(impl, "write %.2f us per span, read %.2f us per span (%d)" % ((t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, n))
