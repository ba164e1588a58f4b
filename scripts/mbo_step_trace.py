"""Prints one outer step of PoissonMBO (between two onehot kernels) from the rocpd database scripts/mbo_step_trace.sh wrote."""
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + '/*.db')[0])
rows = list(db.execute("select name, start, end from kernels order by start"))
try:
    cp = list(db.execute("select name, start, end from memory_copies order by start"))
except Exception:
    cp = []
ev = sorted([(s, e, n) for n, s, e in rows] + [(s, e, 'COPY ' + n) for n, s, e in cp])
idx = [i for i, (s, e, n) in enumerate(ev) if 'onehot' in n.lower()]
a, b = idx[-3], idx[-2]
t0 = ev[a][0]
prev = ''
nsw = 0
for s, e, n in ev[a:b + 1]:
    if 'spmm_sell' in n:
        nsw += 1
        if 'spmm_sell' in prev:
            prev = n
            continue
    print('%9.1f us %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n[:90]))
    prev = n
print('(%d heat sweeps between the first spmm line and the next kernel)' % nsw)
