#!/bin/bash
O=/root/repo/gpurun_out/r03bf
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
cd /tmp
cat > /tmp/big.py <<'PY'
import numpy as np, sys, time
sys.path.insert(0, '/root/repo')
import graphlearning_amd as gl
n = 1000000
g = np.random.default_rng(2)
lab = g.integers(0, 10, size=n); X = (g.normal(size=(10, 64)) * 4)[lab] + g.normal(size=(n, 64))
W = gl.weightmatrix.knn(X, 10)
ti = gl.trainsets.generate(lab, rate=5, seed=0)
m = gl.ssl.poisson(W, solver='gradient_descent')
t0 = time.perf_counter(); p = m.fit_predict(ti, lab[ti]); print('first fit_predict %.3f s' % (time.perf_counter() - t0))
t0 = time.perf_counter(); p = m.fit_predict(ti, lab[ti]); print('second %.3f s' % (time.perf_counter() - t0))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python /tmp/big.py > $O/log.txt 2>&1
grep "fit_predict\|second" $O/log.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import sys, csv
rows = list(csv.DictReader(open(sys.argv[1])))
print('%-70s %7s %10s %10s' % ('kernel', 'calls', 'avg us', 'total ms'))
for r in rows[:34]:
    print('%-70s %7s %10.1f %10.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
