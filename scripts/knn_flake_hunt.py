"""The round-4 flake, hunted where the round-6 soak found it (tests/test_gpu_fuzz.py::test_random_knn_searches_match_ckdtree: "the reordered
search differs", 2 of 36 800 cases with 12 processes on one GPU, 0 of 60 repetitions of the same seeds alone): several processes share the
GPU and repeat searches of random shapes; every process checks BOTH searches of the test against cKDTree and, when one differs, writes
what differed (rows, columns, indices, distances, which of the two searches was wrong, the search's statistics) to gpurun_out/TAG/.
    python scripts/knn_flake_hunt.py TAG PROCESSES SECONDS [seed ...]      (no seeds: random shapes of the test's generator)"""
import os, sys, time, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(seed):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([2, 3, 17, 64, 65, 129, 500, 1500, 3000, 6000]))
    d = int(rng.choice([1, 2, 3, 7, 16, 17, 32, 33, 50, 64, 65, 96, 97, 128, 129, 200, 300]))
    k = int(min(n, rng.choice([1, 2, 5, 11, 12, 13, 21, 28, 29, 40, 60])))
    style = int(rng.integers(0, 4))
    if style == 0:
        X = rng.normal(size=(n, d))
    elif style == 1:
        C = int(rng.integers(2, 12))
        X = rng.normal(size=(C, d))[rng.integers(0, C, size=n)] * 3.0 + rng.normal(size=(n, d))
    elif style == 2:
        X = rng.normal(size=(n, d)) + 50.0
    else:
        X = rng.normal(size=(n, d)) * np.exp(rng.normal(size=(1, d)) * 2.0)
    sim = 'angular' if rng.random() < 0.2 and d > 1 else 'euclidean'
    return X, n, d, k, style, sim


def worker(tag, wid, seconds, seeds):
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    from scipy.spatial import cKDTree
    out = os.path.join(ROOT, 'gpurun_out', tag)
    os.makedirs(out, exist_ok=True)
    log = open(os.path.join(out, 'worker%d.log' % wid), 'a')
    rng = np.random.default_rng(1000 + wid)
    t_end = time.time() + seconds
    ncase = nbad = 0
    cache = {}
    while time.time() < t_end:
        seed = int(seeds[ncase % len(seeds)]) if seeds else int(rng.integers(0, 200000))
        if seed not in cache:
            X, n, d, k, style, sim = case(seed)
            Y = X / np.linalg.norm(X, axis=1)[:, None] if sim == 'angular' else X
            Do, Jo = cKDTree(Y).query(Y, k=k)
            if len(cache) > 64:
                cache.clear()
            cache[seed] = (X, n, d, k, style, sim, Jo.reshape(n, -1), Do.reshape(n, -1))
        X, n, d, k, style, sim, Jo, Do = cache[seed]
        for rep in range(4 if seeds else 1):
            J, D = gl.weightmatrix.knnsearch(X, k, similarity=sim)
            J, D = np.array(J), np.array(D)
            st1 = _hip.knn_stats()
            J2, D2 = _hip.knn_bruteforce(X, k, similarity=sim, want_order=True)
            J2, D2 = np.array(J2), np.array(D2)
            st2 = _hip.knn_stats()
            ncase += 1
            if np.array_equal(J, J2) and np.array_equal(D, D2):
                continue
            nbad += 1
            rows = np.flatnonzero(np.any(J != J2, axis=1) | np.any(D != D2, axis=1))
            scale = max(1.0, float(np.max(Do)))
            rec = {'seed': seed, 'n': n, 'd': d, 'k': k, 'style': style, 'sim': sim, 'rows': rows[:16].tolist(), 'nrows': int(len(rows)),
                   'plain_equals_ckdtree': bool(np.array_equal(J, Jo) and np.max(np.abs(D - Do)) <= 1e-12 * scale),
                   'ordered_equals_ckdtree': bool(np.array_equal(J2, Jo) and np.max(np.abs(D2 - Do)) <= 1e-12 * scale),
                   'stats_plain': {k_: float(v) if not isinstance(v, str) else v for k_, v in st1.items()},
                   'stats_ordered': {k_: float(v) if not isinstance(v, str) else v for k_, v in st2.items()}, 'detail': []}
            for i in rows[:4]:
                cols = np.flatnonzero((J[i] != J2[i]) | (D[i] != D2[i]))
                rec['detail'].append({'row': int(i), 'cols': cols[:12].tolist(), 'J_plain': J[i, cols[:12]].tolist(), 'J_ordered': J2[i, cols[:12]].tolist(),
                                      'J_ckdtree': Jo[i, cols[:12]].tolist(), 'D_plain': D[i, cols[:12]].tolist(), 'D_ordered': D2[i, cols[:12]].tolist(),
                                      'D_ckdtree': Do[i, cols[:12]].tolist(), 'kth_ckdtree': float(Do[i, -1]),
                                      'whole_row_plain': J[i].tolist(), 'whole_row_ordered': J2[i].tolist(), 'whole_row_ckdtree': Jo[i].tolist()})
            log.write(json.dumps(rec) + '\n')
            log.flush()
            os.fsync(log.fileno())
    log.write(json.dumps({'worker': wid, 'cases': ncase, 'mismatches': nbad}) + '\n')
    log.close()


if __name__ == '__main__':
    if sys.argv[1] == '--worker':
        worker(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]), [int(s) for s in sys.argv[5:]])
        sys.exit(0)
    tag, nproc, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    seeds = sys.argv[4:]
    env = dict(os.environ)
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', tag, str(w), str(seconds)] + seeds, env=env) for w in range(nproc)]
    for p in ps:
        p.wait()
    tot = bad = 0
    for w in range(nproc):
        for ln in open(os.path.join(ROOT, 'gpurun_out', tag, 'worker%d.log' % w)):
            j = json.loads(ln)
            if 'worker' in j:
                tot += j['cases']
                bad += j['mismatches']
            else:
                print(json.dumps({k: v for k, v in j.items() if k != 'detail'})[:600])
                for dt in j['detail'][:2]:
                    print('   ', json.dumps({k: v for k, v in dt.items() if not k.startswith('whole')})[:900])
    print('knn_flake_hunt %s: %d processes, %.0f s: %d search pairs, %d mismatches' % (tag, nproc, seconds, tot, bad))
