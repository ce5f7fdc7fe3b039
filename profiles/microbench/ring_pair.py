import sys, time, threading
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import caltech_bifrost_dsp_amd
from caltech_bifrost_dsp_amd import ring as R
impl = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
R.IMPLEMENTATION = impl
r = R.Ring("t", space="system"); r.resize(64, 64 * 8)
gen = r.read(guarantee=True)
cnt = [0]
def reader():
    for iseq in gen:
        for ispan in iseq.read(64):
            cnt[0] += 1
th = threading.Thread(target=reader); th.start()
t0 = time.perf_counter()
with r.begin_writing() as w:
    with w.begin_sequence(time_tag=0, header="{}") as oseq:
        for k in range(N):
            sp = oseq.reserve(64); sp.close()
th.join()
el = time.perf_counter() - t0
print(impl, "%.2f us per gulp (writer+reader), %d read" % (el / N * 1e6, cnt[0]))
