#!/usr/bin/env python3
"""bench.py -- Poisson-learning sweep rate on the MNIST-shaped k=10 graph (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1], SURVEY.md section 8d "Config 2"): n = 70000 vertices,
real MNIST label vector (tests/golden/MNIST_labels.npz), synthetic d = 20 Gaussian-blob
features standing in for the absent MNIST-VAE blob, k = 10 Gaussian-kernel kNN graph built by
the GPU search, one label per class (trainsets.generate(labels, rate=1, seed=0)).  One STEP is
one Poisson gradient-descent solve: u <- D^-1 b + D^-1 W^T u from u = 0 until the reference's
stop test fires (T = 50 sweeps on this graph), all on the device, inputs resident in HBM.
value = sweeps per second (Poisson iters/sec); edges*classes/sec = value * nnz * C is reported
beside it.  N > 1: every rank owns 70000 vertices of an (N*70000)-vertex graph (weak scaling),
vertex-partitioned after an RCM reordering, one RCCL halo exchange per sweep.
"""
import os
import sys
import json
import time
import argparse
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
N_PER_RANK = 70000
D_FEAT = 20
K_NN = 10
N_CLASSES = 10


def load_labels(n_total):
    """MNIST label vector, tiled when the weak-scaling graph is larger than 70000."""
    path = os.path.join(ROOT, 'tests', 'golden', 'MNIST_labels.npz')
    labels = np.load(path)['labels'].astype(np.int64)
    reps = (n_total + len(labels) - 1) // len(labels)
    return np.tile(labels, reps)[:n_total]


def make_features(labels, scale=2.0):
    """SURVEY.md 8d config 2 generator (same as tests/golden/make_golden.py g4_large).  scale = 0.8: the `connected` workload of the
    multi-GPU runs (the config5_hard data: the classes overlap, the kNN graph is ONE component, Poisson accuracy 92 %)."""
    rng = np.random.default_rng(0)
    centers = rng.normal(size=(N_CLASSES, D_FEAT)) * scale
    return centers[labels] + rng.normal(size=(len(labels), D_FEAT))


def config3_data():
    """SURVEY.md 8d config 3 (BASELINE configs[2]): CIFAR label vector (60000), synthetic d = 32 blobs standing in for the absent
    simclr features (same generator as tests/golden/make_golden.py g4_large)."""
    lab = np.load(os.path.join(ROOT, 'tests', 'golden', 'cifar_labels.npz'))['labels'][:60000].astype(np.int64)
    rng = np.random.default_rng(1)
    centers = rng.normal(size=(10, 32)) * 1.2
    return lab, centers[lab] + rng.normal(size=(60000, 32))


def algorithmic_bytes(n, nnz, C, s_v, s_u):
    """Per-sweep algorithmic HBM bytes: CSR values + 4-byte indices, row pointer,
    read u + read Db + write u, and the fused fp64 stop column read + write."""
    return nnz * (s_v + 4) + 4 * (n + 1) + 3 * n * C * s_u + 2 * n * 8


def cpu_baseline(W, train_ind, train_labels, u_hip, T_hip, budget_s=12.0):
    """The oracle (scipy csr_matvecs / csc_matvec, the reference's own arithmetic) timed on this
    host, single-threaded as scipy is: repeated `u = Db + P*u ; v = RW*v` sweeps for ~budget_s."""
    from oracle import gl_oracle as orc
    s = orc.poisson_gd_setup(W, train_ind, train_labels)
    P, Db, RW = s['P'], s['Db'], s['RW']
    n = W.shape[0]
    u_ref, T_ref = orc.poisson_gd(W, train_ind, train_labels, return_T=True)
    parity = bool(T_ref == T_hip and np.array_equal(u_ref, u_hip))
    u = np.zeros((n, s['k']))
    v = s['v0'].copy()
    sweeps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for _ in range(10):
            u = Db + P * u
            v = RW * v
        sweeps += 10
    dt = time.perf_counter() - t0
    return dict(value=sweeps / dt, unit='Poisson iters/sec', cores=1, kind='port',
                sample='%d scipy sweeps (u=Db+P*u; v=RW*v) of the same 70000-vertex graph in %.1f s' % (sweeps, dt),
                edges_classes_per_s=sweeps / dt * P.nnz * s['k']), parity, T_ref


def cpu_baseline_omp(W, train_ind, train_labels, budget_s=4.0):
    """Many-core figure beside the scipy one: the oracle's C loop with the rows of every
    sweep spread over all host cores by OpenMP (oracle/csr_ref.c: ref_poisson_sweeps_omp; bit-identical
    to the scipy sweeps, stop column included)."""
    import ctypes
    from scipy import sparse
    from oracle import gl_oracle as orc
    s = orc.poisson_gd_setup(W, train_ind, train_labels)
    P = sparse.csr_matrix(s['P'])
    Db = np.ascontiguousarray(s['Db'], dtype=np.float64)
    n, C = Db.shape
    lib = orc._c_lib()
    vp = ctypes.c_void_p
    ip, ix, dv = P.indptr.astype(np.int32), P.indices.astype(np.int32), np.ascontiguousarray(P.data)

    def sweeps(T, threads):
        u, tmp = np.zeros((n, C)), np.zeros((n, C))
        w, wt = np.zeros(n), np.zeros(n)
        t0 = time.perf_counter()
        th = lib.ref_poisson_sweeps_omp(ctypes.c_int64(n), ctypes.c_int64(C), ip.ctypes.data_as(vp), ix.ctypes.data_as(vp),
                                        dv.ctypes.data_as(vp), Db.ctypes.data_as(vp), u.ctypes.data_as(vp), tmp.ctypes.data_as(vp),
                                        w.ctypes.data_as(vp), wt.ctypes.data_as(vp), ctypes.c_int64(T), ctypes.c_int(threads))
        return time.perf_counter() - t0, th, (u if T % 2 == 0 else tmp)
    # a 70000-row sweep is small for a big host: more threads is not faster, so the thread count is
    # the best of a short scan (what a user tuning OMP_NUM_THREADS would arrive at)
    ncpu = os.cpu_count() or 1
    best_rate, best_th = 0.0, 1
    for cand in sorted({c for c in (4, 8, 16, 32, 64, 128, ncpu) if c <= ncpu}):
        sweeps(4, cand)                          # thread pool start-up, first touch
        dt, th, _ = sweeps(40, cand)
        if 40 / dt > best_rate:
            best_rate, best_th = 40 / dt, th
    T = int(max(50, min(20000, budget_s * best_rate)))
    dt, th, u = sweeps(T - T % 2, best_th)
    ref = np.zeros((n, C))
    for _ in range(6):
        ref = Db + P * ref
    _, _, u6 = sweeps(6, best_th)
    return dict(value=(T - T % 2) / dt, unit='Poisson iters/sec', cores=int(th), kind='port',
                sample='%d OpenMP sweeps (rows of each sweep over %d threads -- the fastest of a scan up to %d -- stop column fused) in %.1f s' % (T - T % 2, th, ncpu, dt),
                bit_identical_to_scipy=bool(np.array_equal(u6, ref)))


def scale_shard_line(steps=4, T=50, traffic=True):
    """One GPU's share of config 4 (BASELINE.json configs[3]: blobs d=64, k=10, C=10; n = 10^6 vertices per GPU): the
    same sweep kernel at a size where the state (128 MB of records) and the operator (253 MB) no longer sit in the
    L2s, so its rate is an HBM / Infinity-Cache-true one.  Fixed T sweeps per step (min_iter = max_iter)."""
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    n = 1000000
    rng = np.random.default_rng(2)
    labels = rng.integers(0, 10, size=n)
    centers = rng.normal(size=(10, 64)) * 4
    X = centers[labels] + rng.normal(size=(n, 64))
    t0 = time.perf_counter()
    W = gl.weightmatrix.knn(X, K_NN)
    t_graph = time.perf_counter() - t0
    st = _hip.knn_stats()
    del X
    train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
    out = {'n': n, 'nnz': int(W.nnz), 'd': 64, 'sweeps_per_step': T, 'graph_build_s': t_graph,
           'knn_tile_ms': st['tile_ms'], 'knn_filter': st['filter'],
           # useful work 2 n^2 d_padded per second; the bf16x3 filter issues three bf16 MFMAs per product term, so its matrix-pipe
           # utilisation is 3 x this against the 2500 TFLOP/s dense bf16 peak (the f32 filter: this against 157.3)
           # from 2^17 rows on weightmatrix.knn lets the library form cells and skip those that cannot hold a neighbour
           # (glx_knn_clustered: the same lists): only the visited share of the n^2 pairs is contracted, and the tile time
           # includes the cell passes (centres, bounds, sample pre-pass)
           'knn_search': ('cell-pruned, %d cells' % st['cells']) if st['cells'] else 'all pairs',
           'knn_visited_share': st['visited_share'] if st['cells'] else 1.0}
    pairs = float(n) * n * out['knn_visited_share']
    out['knn_tile_tflops'] = 2.0 * pairs * st['dpa'] / st['tile_ms'] / 1e9
    out['knn_mfma_util'] = (3.0 * out['knn_tile_tflops'] / 2500.0) if st['filter'] == 'bf16x3' else out['knn_tile_tflops'] / 157.3
    for dt, dtype, es in (('f64', np.float64, 8), ('f32', np.float32, 4)):
        model = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=(dtype == np.float32), min_iter=T, max_iter=T)
        dev, aux = model._operators()
        source, k = gl.ssl._poisson_source(n, train_ind, labels[train_ind])
        v0 = np.zeros(n)
        v0[train_ind] = 1
        v0 = v0 / np.sum(v0)
        sweep = _hip.Sweep(dev, k, min_iter=T, max_iter=T, use_hipgraph=True)
        sweep.set_problem(aux['D'] * source, v0 / aux['deg'], aux['deg'], aux['vinf'])
        sweep.run()
        ms = 0.0
        l0 = sweep.launches()
        for _ in range(steps):
            ms += sweep.run()[1]
        per = ms * 1e-3 / max(sweep.launches() - l0, 1)
        ab = algorithmic_bytes(n, W.nnz, N_CLASSES, es, es)
        out[dt] = {'avg_launch_us': per * 1e6, 'algorithmic_bytes_per_launch': ab, 'achieved_GBs': ab / per / 1e9,
                   'frac': ab / per / 1e9 / HBM_PEAK_GBS, 'gather_edges_per_s': W.nnz / per}
        if dt == 'f64':
            # what a perfect per-XCD L2 would fetch: every distinct neighbour record once per XCD range and sweep (+ the operator's
            # index / value stream, the row's own bias / stop data, the store of the new iterate), against the counters
            perm = dev.order()
            recs, rec_bytes = distinct_line_bound(W, perm, 128)
            bound = rec_bytes + W.nnz * 12 + 4 * (n + 1) + n * 128 + 2 * n * 8
            out[dt]['distinct_records_over_xcd_ranges'] = recs
            out[dt]['distinct_line_bound_bytes'] = bound
        sweep.close()
        model._cache[1].close()
    if traffic:
        tr, src = measure_traffic(timeout_s=240, child_flag='--traffic-child-scale')
        out['f64']['traffic'] = tr
        out['f64']['traffic_source'] = src
        if tr:
            out['f64']['traffic_over_algorithmic'] = tr / out['f64']['algorithmic_bytes_per_launch']
            out['f64']['traffic_over_distinct_line_bound'] = tr / out['f64']['distinct_line_bound_bytes']
            # what the memory side behind the L2s actually delivers while this kernel runs (counter bytes, not algorithmic ones): the
            # Infinity Cache sits in that path, so this is "beyond-L2" traffic against the HBM peak, an upper estimate of the HBM share
            out['f64']['traffic_GBs'] = tr / (out['f64']['avg_launch_us'] * 1e-6) / 1e9
            # (FETCH_SIZE / WRITE_SIZE count what leaves the L2s: hits of the 256 MB Infinity Cache are in there -- this is NOT an HBM rate;
            # 7 TB/s here exceeds what HBM sustains, 6.3 TB/s.  The share of the fabric reads that goes on to HBM comes from one more pass)
            out['f64']['beyond_l2_traffic_frac_of_hbm_peak'] = out['f64']['traffic_GBs'] / HBM_PEAK_GBS
            share, note = measure_dram_share()
            out['f64']['dram_share_of_fabric_reads'] = share
            out['f64']['dram_share_source'] = note
            if share is not None:
                out['f64']['hbm_read_traffic_frac_of_hbm_peak_estimate'] = out['f64']['traffic_GBs'] * share / HBM_PEAK_GBS
    return out


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def _median_ms(f, device_sync, min_reps=5, min_s=0.25, max_reps=200):
    """Median wall time of f() in ms: every repetition bracketed by device synchronisation, repeated until `min_s` seconds have been
    timed (at least `min_reps` times) -- the headline's rule for the other configurations."""
    ts, total = [], 0.0
    while (total < min_s or len(ts) < min_reps) and len(ts) < max_reps:
        device_sync()
        t0 = time.perf_counter()
        f()
        device_sync()
        ts.append(time.perf_counter() - t0)
        total += ts[-1]
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, len(ts)


def cg_algorithmic_bytes(n, nnz, C, s_v=8, s_u=8):
    """SURVEY.md 8d, CG iteration: SpMM bytes (operator values + 4-byte indices, row pointer, read p, write Ap) plus
    n*C*s_u*(2 [dots: p, Ap] + 3*2 [x, r, p read + write])."""
    return nnz * (s_v + 4) + 4 * (n + 1) + 2 * n * C * s_u + n * C * s_u * 8


def other_configs(W2, labels2, ti2, device_sync, knn_stats2, X2):
    """BASELINE configs[2], configs[4], the default Poisson solver and weightmatrix.knn, each timed like the headline (median of
    synchronised repetitions), each checked against the oracle and / or the digests of the reference's own run
    (tests/golden/g4_large_meta.json, written by tests/golden/make_golden.py from the reference), each with a scipy CPU figure."""
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    from oracle import gl_oracle as orc          # the checker and the CPU baseline only (after the timed regions)
    from scipy import sparse
    meta = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g4_large_meta.json')))
    out = {}
    Cc = N_CLASSES

    # ---- config 3: ssl.laplace (Jacobi-scaled CG on the Dirichlet system), 60 000 vertices, k = 20 --------------------------
    lab3, X3 = config3_data()
    m3 = meta['config3']
    build_ms = _median_ms(lambda: gl.weightmatrix.knn(X3, 20), device_sync, min_reps=3, min_s=0.05)
    W3 = gl.weightmatrix.knn(X3, 20)
    st3 = _hip.knn_stats()
    ti3 = gl.trainsets.generate(lab3, rate=10, seed=0)
    n3 = W3.shape[0]
    nnzA = int(W3.nnz + n3)                           # the Laplacian: W's pattern plus the diagonal
    cgb = cg_algorithmic_bytes(n3, nnzA, Cc)
    c3 = {'workload': 'configs[2]: CIFAR-shaped k=20 kNN graph (n=60000, d=32 synthetic blobs, nnz=%d), ssl.laplace defaults '
                      '(tol=1e-5), trainsets.generate(rate=10, seed=0)' % W3.nnz,
          'graph': {'weightmatrix_knn_ms': build_ms[0], 'knn_tile_ms': st3['tile_ms'], 'nnz': int(W3.nnz),
                    'W_indices_match_reference': _sha(W3.indices.astype(np.int32)) == m3['W_indices_sha']},
          'algorithmic_bytes_per_iteration': cgb}
    # the oracle's solve = the CPU baseline (scipy csr_matvecs inside utils.conjgrad's loop, one core) and the checker
    MAM, Mb, M, idx, F, k = orc.laplace_system(W3, ti3, lab3[ti3])
    t0 = time.perf_counter()
    v_ref, it_ref, _ = orc.conjgrad(MAM, Mb, tol=1e-5, return_iters=True)
    t_cpu = time.perf_counter() - t0
    u_ref = np.zeros((n3, k))
    u_ref[idx, :] = M * v_ref
    u_ref[ti3, :] = F
    for mode in ('default', 'exact'):
        # 'default': ssl.laplace(W) as a user calls it -- reduce='auto', the tolerance-mode reductions handed back to the reference-order
        # ones on long solves or close stop decisions (ssl._solve); 'exact': reference-order reductions, bit-identical iterates
        model = gl.ssl.laplace(W3) if mode == 'default' else gl.ssl.laplace(W3, reduce=mode)
        u = model.fit(ti3, lab3[ti3])
        ms = _median_ms(lambda: model.fit(ti3, lab3[ti3]), device_sync)
        its = int(model.num_iter)
        per_it = ms[0] * 1e-3 / its
        c3[mode] = {'fit_ms': ms[0], 'fit_ms_min': ms[1], 'fit_ms_max': ms[2], 'reps': ms[3], 'cg_iterations': its,
                    'cg_iterations_per_s': its / (ms[0] * 1e-3), 'edges_classes_per_s': nnzA * Cc * its / (ms[0] * 1e-3),
                    'us_per_iteration_of_fit_wall_time': per_it * 1e6,
                    'roofline': {'bound': 'hbm', 'achieved': cgb / per_it / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                 'frac': cgb / per_it / 1e9 / HBM_PEAK_GBS,
                                 'note': 'whole fit (host set-up, uploads, result download included) divided by its iterations'},
                    'parity': {'iterations_equal_oracle': its == int(it_ref),
                               'bit_identical_to_oracle': bool(np.array_equal(u, u_ref)),
                               'max_abs_diff_to_oracle': float(np.max(np.abs(u - u_ref))),
                               'within_1e-5': bool(np.max(np.abs(u - u_ref)) <= 1e-5),
                               'labels_match_reference_run': _sha(model.predict().astype(np.int64)) == m3['pred_sha']}}
    try:       # the same exact-mode fit with the reduction chains walked one dependent addition per row (the form of rounds 1-4; same bits)
        blk3 = model._cache[3].last_block_stats()            # of the 'exact' fit (the loop's last model)
        forms3 = model._cache[3].last_block_forms()
        _hip.CG_EXACT_FORM = 'chain'
        try:
            u_chain3 = model.fit(ti3, lab3[ti3])
            ms_chain3 = _median_ms(lambda: model.fit(ti3, lab3[ti3]), device_sync)
        finally:
            _hip.CG_EXACT_FORM = None
        c3['exact']['reduction_chains'] = {'form': 'blocks of 256 rows', 'blocks_plain_by_record_row_by_row': list(blk3),
                                           'kinds_in_block_form_at_the_end': {'p.Ap': bool(forms3 & 1), 'r.r': bool(forms3 & 2)},
                                           'chain_form_fit_ms': ms_chain3[0], 'speedup_over_chain_form': ms_chain3[0] / c3['exact']['fit_ms'],
                                           'bit_identical_to_chain_form': bool(np.array_equal(u, u_chain3))}
    except Exception as e:          # a measurement beside the line, never the reason there is no line
        c3['exact']['reduction_chains'] = {'error': repr(e)}
    c3['headline'] = 'default'
    c3['default']['reduce'] = 'auto'
    c3['cpu_baseline'] = {'value': it_ref / t_cpu, 'unit': 'CG iterations/s', 'cores': 1, 'kind': 'port',
                          'sample': 'the whole solve: %d iterations of the oracle\'s conjgrad (scipy csr_matvecs) in %.2f s' % (it_ref, t_cpu)}
    out['config3_laplace'] = c3

    # ---- config 2, the DEFAULT Poisson solver (conjugate gradient on the singular normalised system) --------------------------
    m2 = meta['config2']
    model = gl.ssl.poisson(W2)
    u = model.fit(ti2, labels2[ti2])
    ms = _median_ms(lambda: model.fit(ti2, labels2[ti2]), device_sync)
    its = int(model.num_iter)
    n2 = W2.shape[0]
    nnzL = int(W2.nnz + n2)
    cgb2 = cg_algorithmic_bytes(n2, nnzL, Cc)
    t0 = time.perf_counter()
    u_ref2, it_ref2 = orc.poisson_cg(W2, ti2, labels2[ti2], return_iters=True)
    t_cpu2 = time.perf_counter() - t0
    per_it = ms[0] * 1e-3 / its
    # the same fit with the reduction chains walked one dependent addition per row (the form of rounds 1-4; same bits)
    try:
        blk2 = model._cache[1].last_block_stats()
        forms2 = model._cache[1].last_block_forms()
        _hip.CG_EXACT_FORM = 'chain'
        try:
            u_chain = model.fit(ti2, labels2[ti2])
            ms_chain = _median_ms(lambda: model.fit(ti2, labels2[ti2]), device_sync)
        finally:
            _hip.CG_EXACT_FORM = None
        chains2 = {'form': 'blocks of 256 rows, re-decided per kind of reduction during the solve (products that cancel go row by row)',
                   'blocks_plain_by_record_row_by_row': list(blk2), 'kinds_in_block_form_at_the_end': {'p.Ap': bool(forms2 & 1), 'r.r': bool(forms2 & 2)},
                   'chain_form_fit_ms': ms_chain[0], 'chain_form_us_per_iteration': ms_chain[0] * 1e3 / its,
                   'speedup_over_chain_form': ms_chain[0] / ms[0], 'bit_identical_to_chain_form': bool(np.array_equal(u, u_chain))}
    except Exception as e:
        chains2 = {'error': repr(e)}
    out['config2_poisson_cg'] = {
        'workload': 'configs[1] graph, ssl.poisson(W) with its default solver (conjugate_gradient, tol=1e-3): reference-order reductions, '
                    'because the system is singular and the iteration count is part of the contract',
        'fit_ms': ms[0], 'fit_ms_min': ms[1], 'fit_ms_max': ms[2], 'reps': ms[3], 'cg_iterations': its,
        'cg_iterations_per_s': its / (ms[0] * 1e-3), 'edges_classes_per_s': nnzL * Cc * its / (ms[0] * 1e-3),
        'algorithmic_bytes_per_iteration': cgb2,
        'roofline': {'bound': 'hbm', 'achieved': cgb2 / per_it / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': cgb2 / per_it / 1e9 / HBM_PEAK_GBS,
                     'note': 'numpy\'s two row-after-row reduction chains per iteration, walked in block form (csrc/seqsum_exact.h: integer '
                             'block sums confirmed by the exact running sum; same bits as the chain)'},
        'reduction_chains': chains2,
        'parity': {'iterations_equal_oracle': its == int(it_ref2), 'bit_identical_to_oracle': bool(np.array_equal(u, u_ref2)),
                   'iterations_match_reference_run': its == m2['cg_iters'],
                   'labels_match_reference_run': _sha(model.predict().astype(np.int64)) == m2['cg_pred_sha']},
        'cpu_baseline': {'value': it_ref2 / t_cpu2, 'unit': 'CG iterations/s', 'cores': 1, 'kind': 'port',
                         'sample': 'the whole fit of the oracle (operator set-up + %d scipy conjgrad iterations) in %.2f s' % (it_ref2, t_cpu2)}}

    # ---- the default Poisson solver on a CONSISTENT system: the `connected` workload (centre scale 0.8: one component) ---------------
    # config 2's blob graph is ten separate components: its normalised system is singular ten times over and inconsistent in rounding,
    # the reference's own iterate reaches 1e13 along the null space (tests/golden/g4_large_meta.json).  The same solve on a graph that is
    # ONE component (the multi-GPU runs' `connected` workload) is the number for a well-posed Poisson CG; beside it the tolerance-mode
    # reductions (`tree`), which ssl.poisson does not use by default: iterations and max |du| against the reference-order solve.
    try:
        Xc = make_features(labels2, scale=0.8)
        Wc = gl.weightmatrix.knn(Xc, K_NN)
        ncomp = int(sparse.csgraph.connected_components(Wc, directed=False)[0])
        model_c = gl.ssl.poisson(Wc)
        uc = model_c.fit(ti2, labels2[ti2])
        ms_c = _median_ms(lambda: model_c.fit(ti2, labels2[ti2]), device_sync)
        its_c = int(model_c.num_iter)
        nnzLc = int(Wc.nnz + n2)
        cgbc = cg_algorithmic_bytes(n2, nnzLc, Cc)
        t0 = time.perf_counter()
        uc_ref, itc_ref = orc.poisson_cg(Wc, ti2, labels2[ti2], return_iters=True)
        t_cpuc = time.perf_counter() - t0
        per_c = ms_c[0] * 1e-3 / max(its_c, 1)
        dev_c, aux_c = model_c._operators()
        src_c, _k = gl.ssl._poisson_source(n2, ti2, labels2[ti2])
        rhs_c = np.ascontiguousarray(aux_c['D'] * src_c)
        xt, it_t, _e = dev_c.cg(rhs_c, tol=model_c.tol, reduce='tree')
        ms_t = _median_ms(lambda: dev_c.cg(rhs_c, tol=model_c.tol, reduce='tree'), device_sync)
        ut = aux_c['D'] * xt
        blk_c = model_c._cache[1].last_block_stats()
        out['poisson_cg_connected'] = {
            'workload': 'the configs[1] generator with centre scale 0.8 (n=70000, nnz=%d, %d connected component%s), ssl.poisson(W) with its '
                        'default solver (conjugate_gradient, tol=1e-3), trainsets.generate(rate=1, seed=0)' % (Wc.nnz, ncomp, '' if ncomp == 1 else 's'),
            'fit_ms': ms_c[0], 'fit_ms_min': ms_c[1], 'fit_ms_max': ms_c[2], 'reps': ms_c[3], 'cg_iterations': its_c,
            'us_per_iteration_of_fit_wall_time': per_c * 1e6, 'cg_iterations_per_s': its_c / (ms_c[0] * 1e-3),
            'algorithmic_bytes_per_iteration': cgbc,
            'roofline': {'bound': 'hbm', 'achieved': cgbc / per_c / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': cgbc / per_c / 1e9 / HBM_PEAK_GBS},
            'blocks_plain_by_record_row_by_row': list(blk_c),
            'parity': {'iterations_equal_oracle': its_c == int(itc_ref), 'bit_identical_to_oracle': bool(np.array_equal(uc, uc_ref)),
                       'accuracy_percent': float(gl.ssl.ssl_accuracy(model_c.predict(), labels2, ti2))},
            'tolerance_mode_reductions': {'fit_ms': ms_t[0], 'cg_iterations': int(np.max(it_t)), 'max_abs_du_vs_reference_order': float(np.max(np.abs(ut - uc))),
                                          'scale_of_u': float(np.max(np.abs(uc))),
                                          'labels_equal': bool(np.array_equal(np.argmax(ut, axis=1), np.argmax(uc, axis=1))),
                                          'note': 'not what ssl.poisson runs: its contract is the reference\'s iterates bit for bit'},
            'cpu_baseline': {'value': itc_ref / t_cpuc, 'unit': 'CG iterations/s', 'cores': 1, 'kind': 'port',
                             'sample': 'the whole fit of the oracle (operator set-up + %d scipy conjgrad iterations) in %.2f s' % (itc_ref, t_cpuc)}}
        del model_c, Wc, Xc
    except Exception as e:          # a measurement beside the line, never the reason there is no line
        out['poisson_cg_connected'] = {'error': repr(e)}

    # ---- config 5: PoissonMBO on the config-2 graph ---------------------------------------------------------------------------
    m5 = meta['config5']
    priors = gl.utils.class_priors(labels2)
    mbo = gl.ssl.poisson_mbo(W2, priors, solver='gradient_descent', Ns=40, mu=1, T=20)
    prob = mbo.fit(ti2, labels2[ti2])
    ms = _median_ms(lambda: mbo.fit(ti2, labels2[ti2]), device_sync)
    sweeps = int(mbo.poisson_model.num_iter) + 20 * 40
    # CPU: the heat sweeps `u = P*u + Db` are what the reference's 13 s consist of -- a bounded sample of them on this host
    dt = 1 / np.max(orc.degree_vector(W2))
    P = sparse.csr_matrix(sparse.identity(n2) - dt * orc.laplacian(W2))
    ucpu = orc.labels_to_onehot(labels2, Cc).astype(np.float64)
    Db = np.zeros((n2, Cc))
    t0 = time.perf_counter()
    ncpu = 0
    while time.perf_counter() - t0 < 4.0:
        for _ in range(10):
            ucpu = P * ucpu + Db
        ncpu += 10
    t_cpu5 = time.perf_counter() - t0
    out['config5_poisson_mbo'] = {
        'workload': 'configs[4]: poisson_mbo(W, class_priors, solver=gradient_descent, Ns=40, mu=1, T=20) on the config-2 graph',
        'fit_ms': ms[0], 'fit_ms_min': ms[1], 'fit_ms_max': ms[2], 'reps': ms[3], 'sweeps_per_fit': sweeps,
        'sweeps_per_s': sweeps / (ms[0] * 1e-3), 'volume_projections_per_fit': 21,
        'parity': {'labels_match_reference_run': _sha(mbo.predict().astype(np.int64)) == m5['pred_sha'],
                   'prob_matches_reference_run': _sha(np.ascontiguousarray(prob, dtype=np.float64)) == m5['prob_sha']},
        'cpu_baseline': {'value': ncpu / t_cpu5, 'unit': 'heat sweeps/s', 'cores': 1, 'kind': 'port',
                         'sample': '%d scipy sweeps u = P*u + Db of the same operator in %.1f s (the reference\'s fit is 851 of them plus 21 '
                                   'projections)' % (ncpu, t_cpu5)}}

    # ---- stacked gradient-descent trials on the config-2 graph (ssl.ssl_trials: B training sets as column groups of ONE sweep) ----
    out['trials_gd'] = trials_gd_block(W2, labels2, ti2, device_sync)

    # ---- weightmatrix.knn at config 2 ------------------------------------------------------------------------------------------
    ms = _median_ms(lambda: gl.weightmatrix.knn(X2, K_NN), device_sync, min_reps=5, min_s=0.1)
    st = _hip.knn_stats()
    pairs = float(n2) * n2 * (st['visited_share'] if st['cells'] else 1.0)
    # matrix-pipe utilisation by TIME: v_mfma_f32_32x32x16_bf16 occupies a SIMD's matrix pipe for 32 cycles (8 passes of 4);
    # the bf16x3 filter issues three per 32x32x16 product block (two in its concatenated-operand form); 1024 SIMDs at 2.4 GHz
    n_mfma = pairs / (32.0 * 32.0) * (st['dpa'] / 16.0) * ((2.0 if st['concatenated'] else 3.0) if st['filter'] == 'bf16x3' else 1.0)
    util = 32.0 * n_mfma / (1024.0 * st['tile_ms'] * 1e-3 * 2.4e9) if st['filter'] == 'bf16x3' else None
    from scipy.spatial import cKDTree
    t0 = time.perf_counter()
    tree = cKDTree(X2)
    t_tree = time.perf_counter() - t0
    nq = 2000
    t0 = time.perf_counter()
    dq, jq = tree.query(X2[:nq], k=K_NN + 1)
    t_q = time.perf_counter() - t0
    J_gpu, D_gpu = gl.weightmatrix.knnsearch(X2, K_NN + 1)
    out['weightmatrix_knn'] = {
        'workload': 'weightmatrix.knn(X, 10): n=70000, d=20 -> scipy CSR (search, Gaussian weights, symmetrisation)',
        'ms': ms[0], 'ms_min': ms[1], 'ms_max': ms[2], 'reps': ms[3], 'knn_tile_ms': st['tile_ms'], 'knn_total_kernels_ms': st['total_ms'],
        'filter': st['filter'], 'visited_share_of_pairs': (st['visited_share'] if st['cells'] else 1.0),
        'mfma_instructions_estimated': n_mfma, 'matrix_pipe_utilisation_by_time': util,
        'issued_mfma_tflops': 2.0 * 32 * 32 * 16 * n_mfma / (st['tile_ms'] * 1e-3) / 1e12,
        'issued_mfma_frac_of_bf16_peak': (2.0 * 32 * 32 * 16 * n_mfma / (st['tile_ms'] * 1e-3) / 1e12 / 2500.0) if st['filter'] == 'bf16x3' else None,
        'parity': {'first_%d_rows_equal_ckdtree' % nq: bool(np.array_equal(J_gpu[:nq], jq) and np.array_equal(D_gpu[:nq], dq)),
                   'neighbour_lists_match_reference_run': _sha(J_gpu.astype(np.int64)) == m2['J_sha'],
                   'W_indices_match_reference_run': _sha(W2.indices.astype(np.int32)) == m2['W_indices_sha']},
        'cpu_baseline': {'value': nq / t_q, 'unit': 'queries/s', 'cores': 1, 'kind': 'port',
                         'sample': 'cKDTree.query of the first %d of the 70000 points against the full tree (k=11) in %.2f s, tree built in '
                                   '%.2f s; the GPU search answers all 70000 in the `ms` above' % (nq, t_q, t_tree),
                         'gpu_queries_per_s': n2 / (ms[0] * 1e-3)}}
    return out


def grouped_algorithmic_bytes(n, nnz, C, B, s_v=8, s_u=8):
    """SURVEY.md 8d's per-sweep bytes for B training sets as C B columns of one sweep: the operator's values + 4-byte indices and the
    row pointer ONCE, read u + read Db + write u for C B columns, B fp64 stop columns read + written."""
    return nnz * (s_v + 4) + 4 * (n + 1) + 3 * n * C * B * s_u + 2 * n * 8 * B


def trials_gd_block(W, labels, ti0, device_sync, B_head=None, scan_B=(2, 4, 8, 16)):
    """ssl.poisson(solver='gradient_descent') over B training sets at once (glx_sweep_groups; reference ssl.py:292-396 runs them one
    `_fit` at a time): the config-2 training set and published MNIST permutation sets / generated ones as column groups of ONE sweep.
    Timed like the headline: T sweeps per step inside the prepared launch graph, HIP events on the library's stream for the
    per-launch time, median of synchronised repetitions for the wall time; every trial checked against the oracle."""
    import graphlearning_amd as gl
    from graphlearning_amd import ssl as glssl
    from oracle import gl_oracle as orc
    n, nnz, C = W.shape[0], int(W.nnz), N_CLASSES
    B_head = int(B_head or glssl.GD_TRIAL_BATCH)          # the batch ssl_trials uses
    g6 = np.load(os.path.join(ROOT, 'tests', 'golden', 'g6_helpers.npz'))
    pool = [ti0] + [g6['mnist_perm_%d' % i] for i in range(10) if ('mnist_perm_%d' % i) in g6.files]
    pool += [gl.trainsets.generate(labels, rate=1 + s % 3, seed=100 + s) for s in range(24)]
    block = {'workload': 'configs[1] graph, ssl.poisson(solver=gradient_descent) over B training sets as C B = %d B columns of one sweep '
                         '(the config-2 set, the published MNIST permutation sets of tests/golden/g6_helpers.npz, generated sets)' % C,
             'scan': []}
    old = glssl.GD_TRIAL_BATCH
    try:
        for B in sorted(set(tuple(scan_B) + (B_head,))):
            if not glssl._gd_fits(C, B, np.float64):
                continue
            glssl.GD_TRIAL_BATCH = B
            model = gl.ssl.poisson(W, solver='gradient_descent')
            trials = [(t, labels[t]) for t in pool[:B]]
            res = model._fit_batch_device(trials)
            Ts = list(model.num_iter)
            groups = model._operators()[1]['groups']
            for _ in range(3):
                groups.run(used=B)
            l0, dev_ms, reps = groups.launches(), 0.0, 0
            while reps < 10 or dev_ms < 100.0:
                dev_ms += groups.run(used=B)[1]
                reps += 1
            launches = groups.launches() - l0
            per = dev_ms * 1e-3 / max(launches, 1)
            wall = _median_ms(lambda: groups.run(used=B), device_sync, min_reps=5, min_s=0.1)
            ab = grouped_algorithmic_bytes(n, nnz, C, B)
            sweeps = max(Ts)
            entry = {'B': B, 'T': Ts, 'avg_launch_us': per * 1e6, 'step_ms_wall': wall[0], 'sweeps_per_step': sweeps,
                     'trial_sweeps_per_s': B * sweeps / (wall[0] * 1e-3), 'edges_classes_per_s': float(nnz) * C * B * sweeps / (wall[0] * 1e-3),
                     'algorithmic_bytes_per_launch': ab,
                     'roofline': {'bound': 'hbm', 'achieved': ab / per / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ab / per / 1e9 / HBM_PEAK_GBS}}
            if B == B_head:
                ok = True
                for j, (t, tl) in enumerate(trials):
                    u_ref, T_ref = orc.poisson_gd(W, t, tl, return_T=True)
                    # (res was produced by the FIRST run; the timing runs above repeat the same solve into the same state)
                    ok = ok and Ts[j] == T_ref and bool(np.array_equal(np.asarray(groups.fetch(j)), u_ref))
                entry['parity'] = {'every_trial_bit_identical_to_oracle_and_T_equal': bool(ok), 'trials_checked': len(trials)}
                block.update({k: v for k, v in entry.items() if k != 'B'})
                block['B'] = B
            block['scan'].append(entry)
            model._cache[2]['groups'].close()
            model._cache[1].close()
    finally:
        glssl.GD_TRIAL_BATCH = old
    return block


def distinct_line_bound(W, perm, rec_bytes=128, nx=8):
    """What a PERFECT per-XCD L2 would have to fetch per sweep, computed on the host: the rows of the operator are cut into the eight
    contiguous ranges of equal work the plan hands to the XCDs (graph.hip: work of a row = its entries + 3) in the vertex order `perm`
    (perm[new] = caller's row; None = the caller's order); every range must bring in each DISTINCT neighbour record it gathers once.
    Returns (distinct records summed over the ranges, bytes = that many records)."""
    n = W.shape[0]
    lens = np.diff(W.indptr).astype(np.int64)
    if perm is None:
        perm = np.arange(n, dtype=np.int64)
    perm = np.asarray(perm, dtype=np.int64)
    work = np.cumsum(lens[perm] + 3)
    cuts = [0] + [int(np.searchsorted(work, work[-1] * x / nx, side='left')) for x in range(1, nx)] + [n]
    total = 0
    seen = np.zeros(n, dtype=bool)
    for x in range(nx):
        rows = perm[cuts[x]:cuts[x + 1]]
        if len(rows) == 0:
            continue
        seen[:] = False
        # (P = D^-1 W^T of a symmetric W has W's pattern: the rows' neighbour lists are W's)
        rowsel = np.zeros(n, dtype=bool)
        rowsel[rows] = True
        idx = W.indices[np.repeat(rowsel, lens)]
        seen[idx] = True
        total += int(seen.sum())
    return total, total * rec_bytes


def measure_traffic(timeout_s=120, child_flag='--traffic-child', kernel='spmm_sell_kernel<double'):
    """HBM-side bytes per launch of the dominant kernel, MEASURED for this build: two child passes of this script
    under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need separate passes, MI355X_MICROARCH.md), per-dispatch means
    over the sweep kernel's dispatches.  gfx950 correction from the same guide: FETCH_SIZE counts 128-byte requests as
    64 bytes -> doubled.  Returns (bytes or None, source description)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    vals = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='glx_pmc_', dir='/tmp')
        cmd = ['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'run', '--',
               sys.executable, os.path.abspath(__file__), child_flag]
        try:
            # (its own process group: a pass that outlives the limit is ended together with the profiled child, by that group's id)
            proc = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL,
                                    stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                proc.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, '%s pass exceeded %d s' % (ctr, timeout_s)
            fs = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            tot, cnt = 0.0, 0
            for f in fs:
                for r in csv.DictReader(open(f)):
                    if r['Counter_Name'] == ctr and kernel in r['Kernel_Name']:
                        tot += float(r['Counter_Value'])
                        cnt += 1
            if cnt == 0:
                return None, 'no %s samples for the sweep kernel (rocprofv3 pass failed)' % ctr
            vals[ctr] = (tot / cnt, cnt)
        except Exception as e:      # noqa: BLE001 -- the bench line must still be printed
            return None, '%s pass failed: %s' % (ctr, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    traffic = (2.0 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024.0
    measure_traffic.last_child = (child_flag, kernel)
    return traffic, ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes, means over %d / %d dispatches of '
                     'spmm_sell_kernel<double,4,true,false>; (2 x FETCH_SIZE + WRITE_SIZE) KiB' % (vals['FETCH_SIZE'][1], vals['WRITE_SIZE'][1]))


def measure_dram_share(timeout_s=240, child_flag='--traffic-child-scale', kernel='spmm_sell_kernel<double'):
    """Of the read requests the L2s send to the fabric while the sweep kernel runs, the share that goes on to HBM (the rest are hits of the
    Infinity Cache, which FETCH_SIZE counts as well: MI355X_MICROARCH.md, HBM section): one more `rocprofv3 --pmc` child pass over
    TCC_EA0_RDREQ_sum and TCC_EA0_RDREQ_DRAM_sum.  Best effort: (share or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'rocprofv3 not on PATH'
    d = tempfile.mkdtemp(prefix='glx_pmc_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--pmc', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_DRAM_sum', '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'run', '--',
               sys.executable, os.path.abspath(__file__), child_flag]
        proc = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                start_new_session=True)
        try:
            proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            return None, 'pass exceeded %d s' % timeout_s
        tot = {'TCC_EA0_RDREQ_sum': 0.0, 'TCC_EA0_RDREQ_DRAM_sum': 0.0}
        cnt = 0
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] in tot and kernel in r['Kernel_Name']:
                    tot[r['Counter_Name']] += float(r['Counter_Value'])
                    cnt += 1
        if cnt == 0 or tot['TCC_EA0_RDREQ_sum'] <= 0:
            return None, 'no samples of TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_DRAM_sum for the sweep kernel (counters not offered by this rocprofv3?)'
        return tot['TCC_EA0_RDREQ_DRAM_sum'] / tot['TCC_EA0_RDREQ_sum'], 'TCC_EA0_RDREQ_DRAM_sum / TCC_EA0_RDREQ_sum over %d samples' % (cnt // 2)
    except Exception as e:      # noqa: BLE001
        return None, 'pass failed: %s' % e
    finally:
        shutil.rmtree(d, ignore_errors=True)


def scale_graph(n=1000000):
    """One GPU's share of config 4 (blobs d = 64, k = 10, 10 classes, default_rng(2))."""
    import graphlearning_amd as gl
    rng = np.random.default_rng(2)
    labels = rng.integers(0, 10, size=n)
    centers = rng.normal(size=(10, 64)) * 4
    X = centers[labels] + rng.normal(size=(n, 64))
    return gl.weightmatrix.knn(X, K_NN), labels


def traffic_child_scale(n=1000000, T=20):
    """Workload of the counter passes at the shard size: T fp64 sweeps of the n = 10^6 graph, three times."""
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    W, labels = scale_graph(n)
    train_ind = gl.trainsets.generate(labels, rate=5, seed=0)
    model = gl.ssl.poisson(W, solver='gradient_descent', min_iter=T, max_iter=T)
    for _ in range(3):
        model.fit(train_ind, labels[train_ind])


def traffic_child():
    """Workload of the counter passes: the config-2 graph, three steps of the fp64 sweep."""
    import graphlearning_amd as gl
    from graphlearning_amd import _hip
    labels = load_labels(N_PER_RANK)
    W = gl.weightmatrix.knn(make_features(labels), K_NN)
    train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
    model = gl.ssl.poisson(W, solver='gradient_descent')
    for _ in range(3):
        model.fit(train_ind, labels[train_ind])


def run_single(args):
    import graphlearning_amd as gl
    from graphlearning_amd import _hip, _build
    _hip.require_device()
    device_sync = lambda: _hip.check(_hip.load().glx_device_synchronize(), 'glx_device_synchronize')   # no torch in the single-GPU bench
    labels = load_labels(N_PER_RANK)
    X = make_features(labels)
    gl.weightmatrix.knn(X[:4096], K_NN)            # library start-up (HIP context, code objects) is not graph-build time
    # the graph is built four times: the FIRST call at this size also allocates the page-locked result arrays and the device
    # work buffers of its size class (reported as first_call_s); later builds recycle them -- the steady state a user who builds
    # graph after graph sees (knn_plus_weights_s = the fastest of the three)
    build_s = []
    for _ in range(4):
        t0 = time.perf_counter()
        W = gl.weightmatrix.knn(X, K_NN)
        build_s.append(time.perf_counter() - t0)
    t_graph_first, t_graph = build_s[0], min(build_s[1:])
    knn_stats = _hip.knn_stats()
    train_ind = gl.trainsets.generate(labels, rate=1, seed=0)
    train_labels = labels[train_ind]
    # first use of a NEW graph: a fresh matrix, a fresh model, its first fit_predict (operator upload, plan, launch-graph capture,
    # T sweeps, label decision), host arrays in and numpy out -- against `ms_per_step`, which is the sweeps alone
    fresh_ms = []
    for _ in range(4):
        Wf = gl.weightmatrix.knn(X, K_NN)
        t0 = time.perf_counter()
        mf = gl.ssl.poisson(Wf, solver='gradient_descent')
        mf.fit_predict(train_ind, train_labels)
        fresh_ms.append((time.perf_counter() - t0) * 1e3)
        del mf                                   # (the model's teardown -- launch graph, plan, buffers: ~1 ms -- is not part of a first use)
    del Wf
    n, nnz, C = W.shape[0], W.nnz, N_CLASSES

    # the timed region: batches of EXACTLY --steps steps, each bracketed by synchronisation; batches are repeated until
    # >= MIN_TIMED_S of sweeps have been timed (a 20-step batch is 13 ms: one noisy batch must not move the headline);
    # the reported batch is the MEDIAN one, min / max beside it
    MIN_TIMED_S = float(args.min_timed_s)
    results = {}
    for dtype in (np.float64, np.float32):
        model = gl.ssl.poisson(W, solver='gradient_descent', use_cuda=(dtype == np.float32))
        dev, aux = model._operators()
        source, k = gl.ssl._poisson_source(n, train_ind, train_labels)
        Db = aux['D'] * source
        v0 = np.zeros(n)
        v0[train_ind] = 1
        v0 = v0 / np.sum(v0)
        sweep = _hip.Sweep(dev, k, min_iter=model.min_iter, max_iter=model.max_iter, use_hipgraph=True)
        sweep.set_problem(Db, v0 / aux['deg'], aux['deg'], aux['vinf'])
        for _ in range(args.warmup):
            sweep.run()
        batches = []
        T = 0
        total = 0.0
        while total < MIN_TIMED_S or len(batches) < 5:
            device_sync()
            dev_ms = 0.0
            l0 = sweep.launches()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                T, ms = sweep.run()
                dev_ms += ms
            device_sync()
            wall = time.perf_counter() - t0
            batches.append(dict(wall=wall, dev_ms=dev_ms, launches=sweep.launches() - l0))
            total += wall
            if len(batches) >= 2000:
                break
        batches.sort(key=lambda b: b['wall'])
        med = batches[len(batches) // 2]
        u = sweep.fetch()
        results[dtype] = dict(T=T, wall=med['wall'], dev_ms=med['dev_ms'], launches=med['launches'], u=u, info=dev.info(),
                              n_batches=len(batches), wall_min=batches[0]['wall'], wall_max=batches[-1]['wall'],
                              timed_total_s=total)
        sweep.close()

    r64, r32 = results[np.float64], results[np.float32]
    T = r64['T']
    sweeps = args.steps * T
    value = sweeps / r64['wall']
    # dominant kernel: spmm_sell_kernel<double,4,true,false>; HIP events bracket the launches on the library's stream
    abytes = algorithmic_bytes(n, nnz, C, 8, 8)
    avg_launch_s = r64['dev_ms'] * 1e-3 / max(r64['launches'], 1)
    achieved = abytes / avg_launch_s / 1e9
    if args.no_traffic:
        traffic, traffic_source = None, 'skipped (--no-traffic)'
    else:
        traffic, traffic_source = measure_traffic()
    roof = dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                traffic=traffic, traffic_source=traffic_source, libglx_source_hash=_build.source_hash(),
                kernel='spmm_sell_kernel<double,4,true,false>', algorithmic_bytes_per_launch=abytes,
                avg_launch_us=avg_launch_s * 1e6, gather_edges_per_s=nnz / avg_launch_s,
                note='the 70000-vertex working set (9 MB of vertex records + 15 MB of operator image) is L2 / Infinity-Cache '
                     'resident across sweeps, so "HBM" is notional here: the kernel is bound by L2 gather requests (one 128-byte '
                     'record per stored edge).  scale_shard_1e6 below is the same kernel at one GPU\'s share of config 4, where '
                     'the records come from HBM / Infinity Cache; there the bound is the random-line gather rate of the memory '
                     'system (54 G lines/s = 6.9 TB/s of 128-byte lines measured by scripts/probes/gather_probe.hip, '
                     'profiles/r02_gather_probe.txt), reached to ~85-115 %, which caps the algorithmic fraction near 0.2.')
    cpu, parity, T_ref = cpu_baseline(W, train_ind, train_labels, r64['u'], T)
    a32 = algorithmic_bytes(n, nnz, C, 4, 4)
    line = {
        'metric': 'Poisson iters/sec', 'value': value, 'unit': 'iters/s', 'n_gpus': 1, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': r64['wall'] / args.steps * 1e3, 'higher_is_better': True,
        # (one GPU: strong and weak coincide; the label follows the family of lines this one starts: --scaling, default strong = the ONE graph over N GPUs)
        'scaling': getattr(args, 'scaling', 'strong'), 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: MNIST-shaped k=10 kNN graph, n=70000, nnz=%d, C=10, ssl.poisson '
                               'gradient_descent (T=%d sweeps per step, stop test included)' % (nnz, T),
                   'n': n, 'nnz': int(nnz), 'classes': C, 'sweeps_per_step': T},
        'timing': {'batches': r64['n_batches'], 'steps_per_batch': args.steps, 'reported': 'median batch',
                   'ms_per_step_min': r64['wall_min'] / args.steps * 1e3, 'ms_per_step_max': r64['wall_max'] / args.steps * 1e3,
                   'timed_total_s': r64['timed_total_s']},
        'edges_classes_per_sec': value * nnz * C,
        'roofline': roof,
        'cpu_baseline': cpu,
        'cpu_baseline_all_cores': cpu_baseline_omp(W, train_ind, train_labels),
        'speedup_vs_cpu_baseline': value / cpu['value'],
        'parity': {'bit_identical_to_oracle': parity, 'T': T, 'T_oracle': T_ref,
                   'fp32_max_abs_diff': float(np.max(np.abs(r32['u'].astype(np.float64) - r64['u'])))},
        'fp32': {'value': args.steps * r32['T'] / r32['wall'],
                 'roofline_frac': a32 / (r32['dev_ms'] * 1e-3 / max(r32['launches'], 1)) / 1e9 / HBM_PEAK_GBS},
        'graph_build': {'knn_plus_weights_s': t_graph, 'first_call_s': t_graph_first, 'all_calls_s': build_s,
                        'fresh_fit_predict_ms': sorted(fresh_ms[1:])[1], 'fresh_fit_predict_all_ms': fresh_ms,
                        'knn_tile_ms': knn_stats['tile_ms'], 'knn_filter': knn_stats['filter'],
                        'knn_total_ms': knn_stats['total_ms'], 'fallback_rows': knn_stats['fallback_rows'],
                        'sell': r64['info']},
    }
    if not args.no_configs:
        line['configs'] = other_configs(W, labels, train_ind, device_sync, knn_stats, X)
    if not args.no_scale:
        line['scale_shard_1e6'] = scale_shard_line(traffic=not args.no_traffic)
    print(json.dumps(line))


def run_distributed(args):
    from graphlearning_amd import dist_bench
    if args.force_collectives:
        import torch  # noqa: F401  (first: libglx must bind to torch's HIP runtime, see graphlearning_amd/dist.py)
        from graphlearning_amd import dist as gdist
        gdist.FORCE_COLLECTIVES = True
    if args.dist_dry_run:
        dist_bench.main_dry_run(args)
    elif args.config == 4:
        dist_bench.main_config4(args)
    else:
        dist_bench.main(args)


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def supervise_rank(args, argv):
    """What a rank launched by torch.distributed.run does for a multi-GPU line: it is a SUPERVISOR that never touches its GPU.  The
    measurement itself runs in a child process per rank (`bench.py <same arguments> --child`, the ranks' children rendezvous on their own
    port), so that a hang -- a collective that never completes leaves kernels on the device that no later call in the same process
    survives -- costs one ATTEMPT, not the line: the supervisors watch their children in lockstep (one gloo all_reduce of the status
    codes per second), and when any child fails or the attempt's time limit passes, every supervisor kills its child (the device is
    clean again) and the next attempt starts: first the library's own RCCL communicator with captured exchanging sweeps, then the
    torch.distributed engine (eager all_to_all_single).  Supervisor rank 0 relays its child's ONE JSON line to stdout, with the attempts
    that came before it recorded under `attempts`."""
    import signal
    import subprocess
    import datetime
    from graphlearning_amd import dist_bench
    emit = dist_bench._claim_stdout()           # (gloo announces its connections on stdout: the contract is ONE JSON line there)
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    if world != int(args.gpus):
        dist_bench.check_world(args)            # exits with status 2 and the message
    dist.init_process_group('gloo', timeout=datetime.timedelta(minutes=30))
    engines = [args.engine] + (['torch'] if args.engine == 'glx' else [])
    limit = float(os.environ.get('GLX_BENCH_ATTEMPT_S', '600'))
    attempts = []
    final_rc, final_line = 1, None
    for k, engine in enumerate(engines):
        port = [_free_port() if rank == 0 else None]
        dist.broadcast_object_list(port, src=0)
        env = dict(os.environ, MASTER_PORT=str(port[0]), GLX_BENCH_ATTEMPT=str(k), GLX_BENCH_SPAWNED='1' if args.spawned else '0')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # (under torch.distributed.run the ranks are CLIENTS of the launcher's store; the children rendezvous on their own port, where
        # child rank 0 must host the store itself)
        env['TORCHELASTIC_USE_AGENT_STORE'] = 'False'
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != '--spawned'] + ['--child', '--engine', engine]
        t0 = time.perf_counter()
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
        verdict = None
        while verdict is None:
            time.sleep(1.0)
            rc = child.poll()
            code = 0 if rc is None else (1 if rc == 0 else 2)               # running | finished | failed
            late = 1 if time.perf_counter() - t0 > limit else 0
            st = torch.tensor([code, -code, late], dtype=torch.int64)
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            worst, best, late = int(st[0]), -int(st[1]), int(st[2])
            if worst == 2:
                verdict = 'a rank failed'
            elif best == 1 and worst == 1:
                verdict = 'ok'
            elif late:
                verdict = 'no line within %.0f s' % limit
        if verdict != 'ok' and child.poll() is None:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        out = child.communicate()[0] or ''
        lines = [ln for ln in out.splitlines() if ln.startswith('{') and ('"metric"' in ln or '"dry_run"' in ln)]
        if rank == 0:
            ok = verdict == 'ok' and bool(lines)
            attempts.append({'engine': engine, 'outcome': verdict if (verdict != 'ok' or lines) else 'no JSON line', 'seconds': time.perf_counter() - t0})
            if ok:
                final_line = json.loads(lines[-1])
                final_rc = 0
            elif k + 1 < len(engines):
                print('bench.py: attempt %d (engine %s): %s; the children are gone, trying engine %s' % (k, engine, attempts[-1]['outcome'], engines[k + 1]),
                      file=sys.stderr)
        done = [final_rc == 0 if rank == 0 else None]
        dist.broadcast_object_list(done, src=0)
        if done[0]:
            final_rc = 0
            break
    if rank == 0:
        if final_line is not None:
            final_line['attempts'] = attempts
            final_line['supervised'] = True
            emit(json.dumps(final_line))
        else:
            print('bench.py: no attempt produced a line: %s' % attempts, file=sys.stderr)
    dist.barrier()
    dist.destroy_process_group()
    return final_rc


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks here --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same
    arguments>`, one process per GPU -- relay rank 0's ONE JSON line to stdout and return the job's exit status.  The
    ranks themselves refuse to run when the world they find differs from --gpus (dist_bench.check_world), so a
    mis-launched job can never report `n_gpus: 1` under a `--gpus 8` command line."""
    import subprocess
    n = int(args.gpus)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv) + ['--spawned']
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{') and ('"metric"' in ln or '"dry_run"' in ln)]
    if res.returncode != 0 or not lines:
        sys.stdout.write(res.stdout)
        print('bench.py: the %d-rank job failed (exit status %d, %d result lines)' % (n, res.returncode, len(lines)), file=sys.stderr)
        return res.returncode or 1
    line = json.loads(lines[-1])
    if int(line.get('n_gpus', -1)) != n:
        print('bench.py: asked for %d ranks, the job reports %r' % (n, line.get('n_gpus')), file=sys.stderr)
        return 3
    line['launched_by'] = 'bench.py (torch.distributed.run spawned here)'
    print(json.dumps(line))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', type=int, default=2, choices=[2, 4],
                    help='2 (default): BASELINE configs[1], weak scaling over --gpus; 4: configs[3] (blobs d=64), strong scaling of --n vertices')
    ap.add_argument('--n', type=float, default=1e7, help='vertices of --config 4 (default 10^7; kNN alone is ~110 s on one GPU at that size)')
    ap.add_argument('--scaling', default='strong', choices=['strong', 'weak'],
                    help='multi-GPU config 2: strong (default; the stated metric): the ONE 70000-vertex graph over N GPUs; weak: N x 70000 vertices')
    ap.add_argument('--partition', default=None, choices=['cut', 'even', 'cells', 'auto'],
                    help='multi-GPU runs: equal blocks of the locality order (`even`: the default of the strong-scaling line -- every sweep carries '
                         'the halo exchange), contiguous blocks cut between the pieces of the graph (`cut`: the default of --scaling weak and --config 4), '
                         'or the cells of the search assigned to ranks by a balanced partition of their quotient graph')
    ap.add_argument('--no-sides', action='store_true', help='multi-GPU config 2: skip the measurements beside the headline (other partition, `connected` workload)')
    ap.add_argument('--check', action='store_true', help='--config 4: property checks beside the line (counting argument of the search on sampled rows, '
                         'W symmetric / zero diagonal, degree-weighted sums conserved)')
    ap.add_argument('--knn', default='cells', choices=['cells', 'allpairs'], help='--config 4: cell-pruned search (default) or every tile')
    ap.add_argument('--workload', default='blobs', choices=['blobs', 'connected'],
                    help='multi-GPU config 2: the headline features (10 separate clusters) or the same with centre scale 0.8 '
                         '(one connected component: the halo exchange carries real rows)')
    ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 counter passes (roofline.traffic = null)')
    ap.add_argument('--no-scale', action='store_true', help='skip the n = 10^6 shard-size line')
    ap.add_argument('--no-configs', action='store_true', help='skip the configs block (configs 3 and 5, Poisson CG, weightmatrix.knn)')
    ap.add_argument('--min-timed-s', type=float, default=0.5,
                    help='batches of --steps steps are repeated until this many seconds have been timed (at least 5 batches).  Profiling runs '
                         'pass 0: rocprofv3 of ROCm 7.2 segfaults after 16 384 dispatches launched from device graphs (profiles/README.md)')
    ap.add_argument('--engine', default='glx', choices=['glx', 'torch'],
                    help='multi-GPU runs: the library-owned RCCL communicator with captured sweeps (default) or the torch.distributed engine')
    ap.add_argument('--force-dist', action='store_true', help='take the distributed path with one rank (tests)')
    ap.add_argument('--force-collectives', action='store_true', help='distributed path: issue the collectives even with one rank (tests)')
    ap.add_argument('--spawned', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--child', action='store_true', help=argparse.SUPPRESS)           # a rank's measuring process under its supervisor
    ap.add_argument('--test-ops', default=None, help=argparse.SUPPRESS)               # tests: module.py:Class of a CPU stand-in for the rank-local sweep
    ap.add_argument('--test-graph', default=None, help=argparse.SUPPRESS)             # tests: npz with the graph the stand-in sweeps
    ap.add_argument('--traffic-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--traffic-child-scale', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--dist-dry-run', action='store_true',
                    help='start the ranks, rendezvous over gloo, count them, print one line and stop: checks the launch path without a GPU')
    args = ap.parse_args()
    if args.traffic_child:
        traffic_child()
        return
    if args.traffic_child_scale:
        traffic_child_scale()
        return
    launched = 'WORLD_SIZE' in os.environ            # torch.distributed.run (the driver's form for N > 1) or our own spawn
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if launched and not args.child and os.environ.get('GLX_BENCH_SUPERVISE', '1') != '0':
        sys.exit(supervise_rank(args, sys.argv[1:]))     # every launched rank supervises a measuring child (hangs cost an attempt, not the line)
    if (args.gpus > 1 or args.config == 4 or args.dist_dry_run or int(os.environ.get('WORLD_SIZE', '1')) > 1
            or args.force_dist):
        run_distributed(args)
    else:
        run_single(args)


if __name__ == '__main__':
    main()
