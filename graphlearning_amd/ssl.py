"""Graph-based semi-supervised learning on the MI355X: the learners of reference
graphlearning/ssl.py named by the north star -- `poisson` (:513-693), `poisson_mbo`
(:695-839), `laplace` (:1106-1261) -- behind the reference's `ssl` contract (:131-510):
`model = poisson(W, ...)`, `model.fit(train_ind, train_labels) -> (n,C)`,
`model.predict()`, `model.fit_predict(...)`.

As in the reference's own device seam (`if self.use_cuda:` ssl.py:649-663, :807-823) the
host prepares the sparse operator and the dense right-hand side with scipy (O(nnz), once
per fit) and the device runs every iteration; unlike the reference, the stop test, the
conjugate-gradient solver and the volume-constrained projection run on the device too.

Precision: `use_cuda=False` (default) computes in fp64 -- iterates are bit-identical to the
reference CPU path for the sweeps, within rounding for CG; `use_cuda=True` computes in
fp32 like the reference's torch.sparse.addmm branch and returns float32.
"""
import os
import sys
import numpy as np
from scipy import sparse
from . import graph as graph_mod
from . import utils
from . import _hip


results_dir = os.path.join(os.getcwd(), 'results')


class _DeviceState:
    """Result of a fit that still lives in a prepared sweep's device buffer: `prob` is read back only when somebody
    looks at it, label decisions (predict / volume_label_projection) run on the device state directly."""

    def __init__(self, sweep):
        self.sweep = sweep
        self.generation = getattr(sweep, 'generation', 0)

    def valid(self):
        return self.sweep._h.value and getattr(self.sweep, 'generation', 0) == self.generation

    def fetch(self):
        if not self.valid():
            raise RuntimeError('the device state of this fit has been overwritten by a later solve')
        return self.sweep.fetch()


class ssl:
    """Base class, reference ssl.py:131-510."""

    # `prob` (the (n, C) result of the last fit, reference ssl.py:146) is an ordinary attribute for callers; when a fit
    # leaves its result on the device it is read back on first access
    @property
    def prob(self):
        if self._prob is None and self._dev_state is not None:
            self._prob = self._dev_state.fetch()
        return self._prob

    @prob.setter
    def prob(self, value):
        self._prob = value
        self._dev_state = None

    def _set_result(self, res):
        if isinstance(res, _DeviceState):
            self._prob = None
            self._dev_state = res
        else:
            self.prob = res

    def _device_state(self):
        """The sweep holding the current `prob` on the device, if it is still there."""
        st = self._dev_state
        return st if (st is not None and st.valid()) else None

    def __init__(self, W, class_priors):
        self._prob = None
        self._dev_state = None
        if W is None:
            self.graph = None
        else:
            self.set_graph(W)
        self.prob = None
        self.fitted = False
        self.name = ''
        self.accuracy_filename = ''
        self.requires_eig = False
        self.onevsrest = False
        self.similarity = True
        self.class_priors = class_priors
        if self.class_priors is not None:
            self.class_priors = self.class_priors / np.sum(self.class_priors)
        self.weights = 1
        self.class_priors_error = 1
        self.device = None      # None -> _hip.default_device() (set_default_device(LOCAL_RANK) in multi-GPU processes)

    def set_graph(self, W):
        st = getattr(self, '_dev_state', None)
        if st is not None and getattr(self, '_prob', None) is None and st.valid():
            self._prob = st.fetch()          # a result still on the device belongs to the old graph's sweep: read it back first
            self._dev_state = None
        if type(W) == graph_mod.graph:
            self.graph = W
        else:
            self.graph = graph_mod.graph(W)
        self._cache = None      # device-resident operators belong to the previous graph

    def _graph_key(self):
        """Identity of the graph the cached device operators were built from: the CONTENT fingerprint of the weight matrix
        (utils.matrix_fingerprint), not its address -- a matrix edited in place between two fits is a new graph, as it is for
        the reference, which rebuilds its operators in every fit (ssl.py:615-644).  Inside one ssl_trials loop the matrix cannot
        change under us, so the fingerprint of the loop's first fit is reused (`_trusted_key`)."""
        W = self.graph.weight_matrix
        trusted = getattr(self, '_trusted_key', None)
        if trusted is not None and trusted[0] is W:
            return trusted[1]
        key = utils.matrix_fingerprint(W)
        if trusted is not None:
            self._trusted_key = (W, key)
        self._last_key = (W, key)
        return key

    def _speculate_key(self):
        """A fit on the matrix OBJECT the previous fit saw starts on the device operators cached under that fit's fingerprint
        while a helper thread hashes the matrix again (0.25-0.3 ms for 14-25 MB, a quarter of a config-2 fit); `_confirm_key` then
        says whether the content really was unchanged -- if not, the fit is repeated on operators rebuilt from the edited
        matrix, so an in-place edit between two fits is honoured exactly as before.  Returns the pending check or None."""
        last = getattr(self, '_last_key', None)
        W = self.graph.weight_matrix
        if not SPECULATIVE_FITS or last is None or last[0] is not W or getattr(self, '_trusted_key', None) is not None:
            return None
        fut = _hash_pool().submit(utils.matrix_fingerprint, W)
        self._trusted_key = (W, last[1])
        return (W, last[1], fut)

    _MUTABLE = ('weights', 'class_priors_error', 'num_iter', 'stop_settled')

    def _mutable_state(self):
        """What a `_fit` may change in the model besides its result (and in the inner Poisson model of PoissonMBO)."""
        def snap(obj):
            return {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in
                    ((k, getattr(obj, k)) for k in self._MUTABLE if hasattr(obj, k))}
        inner = getattr(self, 'poisson_model', None)
        return snap(self), (snap(inner) if inner is not None else None)

    def _restore_state(self, saved):
        if saved is None:
            return
        for k, v in saved[0].items():
            setattr(self, k, v)
        inner = getattr(self, 'poisson_model', None)
        if inner is not None and saved[1] is not None:
            for k, v in saved[1].items():
                setattr(inner, k, v)

    def _confirm_key(self, pending):
        W, assumed, fut = pending
        self._trusted_key = None
        real = fut.result()
        self._last_key = (W, real)
        return real == assumed

    def volume_label_projection(self):
        """Volume-constrained label decision (reference ssl.py:172-209) on the device:
        at most 1e4 steps of w += -0.1 (class_size - priors); w /= w[0], stop at max error
        <= 1e-3.  Updates self.weights / self.class_priors_error; returns the labels."""
        st = self._device_state()
        k = st.sweep.C if st is not None else self.prob.shape[1]
        w = np.ones((k,)) if type(self.weights) == int else self.weights
        if st is not None:
            labels, w, err, _ = st.sweep.project(self.class_priors, w, max_steps=10000, similarity=self.similarity)
        else:
            labels, w, err, _ = _hip.argmax_project(self.prob, self.class_priors, w, max_steps=10000,
                                                    similarity=self.similarity, device=self.device)
        self.weights = w
        self.class_priors_error = err
        return labels

    def get_accuracy_filename(self):
        fname = self.accuracy_filename
        if self.class_priors is not None:
            fname += '_classpriors'
        fname += '_accuracy.csv'
        return fname

    def predict(self, ignore_class_priors=False):
        """argmax of the globally min/max-normalised scores times the class weights
        (reference ssl.py:230-266), on the device."""
        if self.fitted == False:
            sys.exit('Model has not been fitted yet.')
        st = self._device_state()
        k = st.sweep.C if st is not None else self.prob.shape[1]
        if ignore_class_priors or type(self.weights) == int:
            w = np.ones((k,))
        else:
            w = self.weights
        if st is not None:      # the fit's result is still on the device: decide there, only the labels come back
            labels, _, _, _ = st.sweep.project(None, w, max_steps=0, similarity=self.similarity)
            return labels
        labels, _, _, _ = _hip.argmax_project(self.prob, None, w, max_steps=0, similarity=self.similarity,
                                              device=self.device)
        return labels

    def fit_predict(self, train_ind, train_labels, all_labels=None):
        self._fit_only(train_ind, train_labels, all_labels=all_labels)
        return self.predict()

    def fit(self, train_ind, train_labels, all_labels=None):
        """reference ssl.py:439-481."""
        self._fit_only(train_ind, train_labels, all_labels=all_labels)
        return self.prob

    def _fit_only(self, train_ind, train_labels, all_labels=None):
        """`fit` without reading the result back: `prob` stays on the device until it is looked at."""
        if self.graph is None:
            sys.exit('SSL object has no graph. Use graph.set_graph() to provide a graph for SSL.')
        self.fitted = True
        train_ind = np.asarray(train_ind)
        train_labels = np.asarray(train_labels)
        if self.onevsrest:
            unique_labels = np.unique(train_labels)
            self.prob = np.zeros((self.graph.num_nodes, len(unique_labels)))
            for i, l in enumerate(unique_labels):
                self.prob[:, i] = self._fit(train_ind, train_labels == l)
        else:
            pending = self._speculate_key()
            # a speculative fit runs on operators cached for the matrix's PREVIOUS content: whatever it changes in the model
            # (PoissonMBO's volume weights, iteration counts) is put back before the fit is repeated on the edited matrix, and an
            # exception it raises only counts if the content really was unchanged
            saved = self._mutable_state() if pending is not None else None
            try:
                res = self._fit(train_ind, train_labels, all_labels=all_labels)
            except BaseException as exc:
                if pending is None:
                    raise
                if not isinstance(exc, Exception):
                    self._trusted_key = None
                    raise
                if self._confirm_key(pending):      # the content was what the cached operators were built from: a real failure
                    raise
                res, pending = None, None
                self._restore_state(saved)
                res = self._fit(train_ind, train_labels, all_labels=all_labels)
            if pending is not None and not self._confirm_key(pending):
                self._restore_state(saved)
                res = self._fit(train_ind, train_labels, all_labels=all_labels)     # the matrix was edited in place: operators from its new content
            self._set_result(res)
        if self.class_priors is not None:
            self.volume_label_projection()

    def ssl_trials(self, trainsets, labels, num_cores=1, tag='', save_results=True, overwrite=False, num_trials=-1):
        """Run the learner on a list of training sets and record `Number of labels,Accuracy[,...]`
        rows to results/<tag><accuracy filename> (reference ssl.py:292-396, same file format,
        same abort-if-exists rule).  Trials run one after another on the GPU with the operator
        resident on the device (stacked as column groups of one solve where the learner supports it).
        `num_cores` is the reference's number of joblib worker PROCESSES (ssl.py:390-396); here the parallel axis is
        one process per GPU: with num_cores > 1 inside an initialised torch.distributed job of more than one rank the
        trials are shared out over the ranks (dist.ssl_trials_distributed: every rank its own GPU, rank 0 writes the file);
        in a single process the argument changes nothing -- one GPU runs the trials faster than the reference's workers."""
        if num_cores > 1:
            try:
                import torch.distributed as tdist
                multi = tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1
            except ImportError:
                multi = False
            if multi:
                from . import dist as gdist
                # (a collective call: every rank of the job must reach it -- said out loud on every rank, because the reference's
                # num_cores means joblib workers of ONE process and a ported script may call this on rank 0 only)
                print('ssl_trials(num_cores=%d): sharing the trials over the %d ranks of the torch.distributed job (rank %d); '
                      'every rank must call ssl_trials -- pass num_cores=1 to run them all here'
                      % (num_cores, tdist.get_world_size(), tdist.get_rank()), file=sys.stderr, flush=True)
                gdist.ssl_trials_distributed(self, trainsets, labels, tdist, tag=tag, save_results=save_results, overwrite=overwrite,
                                             num_trials=num_trials)
                return
        if num_trials > 0:
            trainsets = trainsets[:num_trials]
        print('\nModel: ' + self.name)
        outfile = None
        with_priors = self.class_priors is not None
        header = 'Number of labels,Accuracy,Accuracy with class priors,Class priors error' if with_priors else 'Number of labels,Accuracy'
        if save_results:
            if not os.path.exists(results_dir):
                os.makedirs(results_dir)
            outfile = os.path.join(results_dir, tag + self.get_accuracy_filename())
            if (not overwrite) and os.path.exists(outfile):
                print('Aborting: SSL trial (' + self.get_accuracy_filename() + ') already completed , and overwrite is False.')
                return
            with open(outfile, 'w') as f:
                f.write(header + '\n')
            print('Results File: ' + outfile)
        print('\n' + header)
        for row in self._trial_rows(trainsets, labels):
            print(row)
            if save_results:
                with open(outfile, 'a+') as f:
                    f.write(row + '\n')

    def _trial_rows(self, trainsets, labels):
        """The result row (reference ssl.py:339-343, 381-388) of every training set, in order."""
        with_priors = self.class_priors is not None
        labels = np.asarray(labels)
        trainsets = [np.asarray(t) for t in trainsets]
        batch = 1 if self.onevsrest else max(1, int(self._trial_batch_size(labels)))
        self._trusted_key = (None, None)         # the graph is fingerprinted once for the whole loop (_graph_key)
        inner = getattr(self, 'poisson_model', None)
        if inner is not None:
            inner._trusted_key = (None, None)
        try:
            yield from self._trial_rows_loop(trainsets, labels, batch, with_priors)
        finally:
            self._trusted_key = None
            if inner is not None:
                inner._trusted_key = None

    def _trial_rows_loop(self, trainsets, labels, batch, with_priors):
        for pos in range(0, len(trainsets), batch):
            group = trainsets[pos:pos + batch]
            # trials that share the graph are stacked as extra right-hand-side columns of ONE device
            # solve where the learner supports it (column for column the same result as one by one)
            probs = self._fit_batch_device([(t, labels[t]) for t in group]) if len(group) > 1 else None
            for j, train_ind in enumerate(group):
                if probs is None:
                    pred = self.fit_predict(train_ind, labels[train_ind])
                else:
                    self.fitted = True
                    self._set_result(probs[j])
                    if self.class_priors is not None:
                        self.volume_label_projection()
                    pred = self.predict()
                accuracy = ssl_accuracy(pred, labels, train_ind)
                if with_priors:
                    plain = ssl_accuracy(self.predict(ignore_class_priors=True), labels, train_ind)
                    yield '%d,%.2f,%.2f,%.5f' % (len(train_ind), plain, accuracy, self.class_priors_error)
                else:
                    yield '%d' % len(train_ind) + ',%.2f' % accuracy

    def _trial_batch_size(self, labels):
        """How many trials ssl_trials hands to _fit_batch at once (1 = one by one)."""
        return 1

    def _fit_batch(self, trials):
        """Fit several (train_ind, train_labels) pairs on the same graph in one device call and
        return their (n, C) results, or None when the learner has no batched path."""
        return None

    def _fit_batch_device(self, trials):
        """_fit_batch whose results may stay on the device (_DeviceState entries, valid until the learner's next solve):
        what ssl_trials consumes -- it only needs the label decision of every trial."""
        return self._fit_batch(trials)

    def trials_statistics(self, tag=''):
        """Mean / standard deviation of the accuracies recorded by ssl_trials, per label rate
        (reference ssl.py:398-436)."""
        X = np.loadtxt(os.path.join(results_dir, tag + self.get_accuracy_filename()), delimiter=',', skiprows=1, ndmin=2)
        num_train = np.unique(X[:, 0])
        acc_mean, acc_stddev = [], []
        for m in num_train:
            Y = X[X[:, 0] == m, 1:]
            acc_mean += [np.mean(Y, axis=0)]
            acc_stddev += [np.std(Y, axis=0)]
        return num_train, np.array(acc_mean), np.array(acc_stddev), int(len(X[:, 0]) / len(num_train))

    def _fit(self, train_ind, train_labels, all_labels=None):
        raise NotImplementedError('Must override _fit')


def _free_order(W, n):
    """The vertex order that came with the graph for free: the cell order of the search that built W (weightmatrix.knn stamps it
    on its result).  Operators that are used for a handful of CG solves take it when it is there and keep the caller's order
    when it is not -- the library's own pass over the graph (O(nnz) on the host) would cost more than such a solve gains."""
    order = getattr(W, '_glx_order', None)
    return order if (order is not None and len(order) == n) else None


# relative half-width around 1/n inside which a fused stop value is re-derived with the reference's recurrence
# (poisson._settle_stop); the fused and the reference values differ by <= 1e-13 relative
STOP_BAND = 1e-9


# Training sets per stacked gradient-descent sweep (poisson._fit_batch_gd): whole 128-byte lines per gather (ten fp64 columns alone
# fill 80 of a line's 128 bytes) and the operator's index / value stream, launch and prologue paid once per batch.  Measured at
# config 2 (70 000 vertices, 10 classes; profiles/r05_trials_gd.txt): 8.9 us per trial and sweep at 4 trials per record (35.3 us per
# launch), 9.5 at 8, 11.6 at 2, 12.4 for a single fit's sweep -- an XCD's share of the stacked state outgrows its 4 MB L2 from 4
# trials on (L2 hit rate 71 % at 4, 53 % at 8), and what the lines gain the misses take back.
GD_TRIAL_BATCH = 4


def _gd_fits(k, B, dtype):
    lanes = (B * k + 3) // 4 + ((B + 3) // 4 if np.dtype(dtype) == np.float64 else (B + 1) // 2)
    return 2 <= B <= 32 and lanes <= 64


def _gd_batch(k, dtype):
    B = GD_TRIAL_BATCH
    while B > 1 and not _gd_fits(k, B, dtype):
        B -= 1
    return max(B, 1)


_HASH_POOL = None
# False: a fit on a matrix object the previous fit saw hashes the matrix FIRST and starts afterwards (the fit never runs on a guess).
# Results are identical either way (tests/test_gpu_switches.py); the switch is the off position of the speculation for ablation runs.
SPECULATIVE_FITS = True


def _hash_pool():
    """One helper thread for the fingerprint that runs beside a fit (created on first use, dropped in a forked child: threads do
    not survive fork)."""
    global _HASH_POOL
    if _HASH_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _HASH_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix='glx-fingerprint')
    return _HASH_POOL


def _drop_hash_pool():
    global _HASH_POOL
    _HASH_POOL = None


if hasattr(os, 'register_at_fork'):
    os.register_at_fork(after_in_child=_drop_hash_pool)


def _poisson_operator_symmetric(W):
    """deg, D^-1 and P = D^-1 W^T (reference ssl.py:615-617, 634-635, 642) for a W known to be symmetric bit for bit with an
    empty diagonal (weightmatrix.knn's output): row i of W^T is row i of W, so nothing is transposed.  scipy's `D * W.transpose()`
    turns W^T into a sorted CSR and multiplies with csr_matmat, which emits every row's entries in REVERSE order (the linked list
    it builds is walked from the last insertion); the same arrays are written down directly: row i = (dinv_i * w_ij) for j
    descending.  Identical to the scipy expressions entry for entry (tests/test_host_logic.py)."""
    n = W.shape[0]
    plain = (W.indptr.dtype == np.int32 and W.indices.dtype == np.int32 and W.data.dtype == np.float64
             and W.data.flags.c_contiguous and W.indices.flags.c_contiguous and W.indptr.flags.c_contiguous)
    if plain and _hip.load(required=False) is not None:
        # the same arrays by plain loops in the library (host code): the numpy formulation below costs 8 ms at 70 000 vertices
        deg = _hip.host_row_sums(W)
        dinv = deg ** (-1)
        indices, data = _hip.host_reverse_scale_rows(W, dinv)
        P = sparse.csr_matrix((data, indices, W.indptr.copy()), shape=(n, n))
        P.has_sorted_indices = False
        return P, deg, dinv
    deg = W * np.ones(n)                                   # graph.degree_vector (W - spdiags(diag) == W: no diagonal entries)
    dinv = deg ** (-1)                                     # graph.degree_matrix(p=-1): d ** p
    indptr = W.indptr
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
    rev = indptr[rows].astype(np.int64) + indptr[rows + 1] - 1 - np.arange(W.nnz, dtype=np.int64)    # mirror inside the row
    P = sparse.csr_matrix(((dinv[rows] * W.data)[rev], W.indices[rev], indptr.copy()), shape=(n, n))
    P.has_sorted_indices = False
    return P, deg, dinv


def _poisson_source(n, train_ind, train_labels):
    """b[train] = onehot - mean(onehot) (reference ssl.py:619-622)."""
    k = len(np.unique(train_labels))
    onehot = utils.labels_to_onehot(train_labels, k)
    source = np.zeros((n, onehot.shape[1]))
    source[train_ind] = onehot - np.mean(onehot, axis=0)
    return source, k


class poisson(ssl):
    def __init__(self, W=None, class_priors=None, solver='conjugate_gradient', p=1, use_cuda=False, min_iter=50,
                 max_iter=1000, tol=1e-3, spectral_cutoff=10):
        """Poisson learning, reference ssl.py:513-693.  Solvers 'conjugate_gradient'
        (default) and 'gradient_descent' run on the GPU; 'spectral' (an eigensolver) is out
        of this package's scope."""
        super().__init__(W, class_priors)
        if solver not in ['conjugate_gradient', 'spectral', 'gradient_descent']:
            sys.exit('Invalid Poisson solver')
        self.solver = solver
        self.p = p
        if p != 1:
            self.solver = 'spectral'
        self.use_cuda = use_cuda
        self.min_iter = min_iter
        self.max_iter = max_iter
        self.tol = tol
        self.spectral_cutoff = spectral_cutoff
        fname = '_poisson'
        if self.p != 1:
            fname += '_p%.2f' % p
        if self.solver == 'spectral':
            fname += '_N%d' % self.spectral_cutoff
            self.requries_eig = True
        self.accuracy_filename = fname
        self.name = 'Poisson Learning'
        self.num_iter = None        # sweeps / CG iterations of the last fit
        self.stop_settled = None    # (T fused, T reference recurrence) when the last GD fit had a stop value within STOP_BAND of 1/n

    def _dtype(self):
        # the reference's use_cuda switch only touches the gradient-descent branch (ssl.py:649)
        return np.float32 if (self.use_cuda and self.solver == 'gradient_descent') else np.float64

    def _operators(self):
        """Host-side setup shared by every fit on this graph (reference ssl.py:615-617,
        626-627, 634-635, 642-644): zero the diagonal, degrees, P = D^-1 W^T or the
        normalised Laplacian; uploaded once and kept on the device."""
        fp = self._graph_key()
        key = (fp, self.solver, self._dtype())
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1], self._cache[2]
        n = self.graph.num_nodes
        W = self.graph.weight_matrix
        # (a stamped matrix is weightmatrix.knn's output, unchanged: symmetric bit for bit, no diagonal, no stored zeros)
        fast = self.solver != 'conjugate_gradient' and utils.known_symmetric(W, fp)
        if not fast:
            W = W - sparse.spdiags(W.diagonal(), 0, n, n)
            G = graph_mod.graph(W)
        aux = {}
        if self.solver == 'conjugate_gradient':
            L = G.laplacian(normalization='normalized')
            aux['D'] = G.degree_matrix(p=-0.5)
            # one-shot CG solves spend their time in the reference-order reductions, not in the SpMM: the library's locality
            # renumbering (an O(nnz) host pass) would not pay for itself -- the search's cell order, when W carries it, is free
            order = _free_order(self.graph.weight_matrix, n)
            dev = _hip.DeviceGraph(L, dtype=self._dtype(), device=self.device, keep_order=order is None, order=order)
        elif fast:
            order = getattr(W, '_glx_order', None) if (self._dtype() == np.float64 or n < (1 << 17)) else None
            if order is not None and len(order) != n:
                order = None
            plain = (W.indptr.dtype == np.int32 and W.indices.dtype == np.int32 and W.data.dtype == np.float64)
            if plain:
                # W goes up as it is (its arrays are page-locked when weightmatrix.knn made them) and stays on the device; degrees
                # there; P = D^-1 W^T is formed while the sliced-ELL image is filled (reversed rows times 1 / degree, the entries
                # scipy's `D * W.transpose()` writes down: _poisson_operator_symmetric is the host form of the same arrays)
                dev, deg = _hip.DeviceGraph.resident(W, dtype=self._dtype(), device=self.device, order=order, want_row_sums=True)
                dinv = deg ** (-1)
                dev.set_row_transform(dinv, reverse_rows=True)
            else:
                P, deg, dinv = _poisson_operator_symmetric(W)
                dev = None
            aux['D'] = sparse.spdiags(dinv, 0, n, n).tocsr()
            aux['dinv'] = dinv
            aux['zero_degree'] = bool(np.any(~np.isfinite(aux['dinv'])))
            aux['deg'] = deg
            aux['vinf'] = deg / np.sum(deg)
            # the cell order of the search that built W, if it was a clustered one: for the fp64 sweep as good as the library's own
            # pass over the graph (251 vs 250 us at 10^6 vertices; 12.45 vs 12.59 us at 70 000) and free (0.13 s / 3.7 ms); the fp32
            # sweep at 10^6 vertices is 4 % faster on the library's order (201 vs 209 us), so that mode keeps paying for it there
            if dev is None:
                dev = _hip.DeviceGraph(P, dtype=self._dtype(), device=self.device, order=order)
        else:
            D = G.degree_matrix(p=-1)
            P = D * W.transpose()
            deg = G.degree_vector()
            aux['D'] = D
            aux['dinv'] = D.diagonal()
            aux['zero_degree'] = bool(np.any(~np.isfinite(aux['dinv'])))
            aux['deg'] = deg
            aux['vinf'] = deg / np.sum(deg)
            dev = _hip.DeviceGraph(P, dtype=self._dtype(), device=self.device)
        if self._cache is not None:
            if self._cache[2].get('sweep') is not None:
                self._cache[2]['sweep'].close()
            if self._cache[2].get('groups') is not None:
                self._cache[2]['groups'].close()
            self._cache[1].close()
        self._cache = (key, dev, aux)
        return dev, aux

    def _fit(self, train_ind, train_labels, all_labels=None):
        n = self.graph.num_nodes
        if self.solver == 'conjugate_gradient':       # reference ssl.py:624-629
            source, k = _poisson_source(n, train_ind, train_labels)
            dev, aux = self._operators()
            D = aux['D']
            x, it, _ = dev.cg(np.ascontiguousarray(D * source, dtype=self._dtype()), tol=self.tol)
            self.num_iter = it
            u = D * x
        elif self.solver == 'gradient_descent':       # reference ssl.py:631-677
            dev, aux = self._operators()
            train_ind = np.asarray(train_ind)
            k = len(np.unique(train_labels))
            # prepared sweep (device buffers + captured launch graph) kept with the operator: repeated
            # fits on one graph (ssl_trials) only upload the new right-hand side
            key = (k, self.min_iter, self.max_iter)
            if aux.get('sweep_key') != key:
                if aux.get('sweep') is not None:
                    aux['sweep'].close()
                aux['sweep'] = None
                if self.max_iter > 0:
                    aux['sweep'] = _hip.Sweep(dev, k, min_iter=self.min_iter, max_iter=self.max_iter, use_hipgraph=True)
                    aux['sweep'].set_vectors(aux['deg'], aux['vinf'])
                aux['sweep_key'] = key
            Db = None
            if aux['sweep'] is None:
                u, T = np.zeros((n, k), dtype=self._dtype()), 0
            elif (len(np.unique(train_ind)) == len(train_ind) and not aux['zero_degree']
                  and (len(train_ind) == 0 or (train_ind.min() >= 0 and train_ind.max() < n))):
                # (numpy-style negative indices, which `source[train_ind] = ...` of the reference accepts, take the dense branch below)
                # Db = D*source and v = 1_train/m are nonzero on the labelled rows only (ssl.py:620-622, 636, 639-641):
                # those m rows are all a new training set uploads; deg and vinf went up with the prepared sweep
                onehot = utils.labels_to_onehot(train_labels, k)
                Db_rows = aux['dinv'][train_ind, None] * (onehot - np.mean(onehot, axis=0))   # rows of D*source: one product per entry
                vval = 1.0 / float(len(train_ind))                        # v[train] = 1; v = v/np.sum(v)
                w0_rows = vval / aux['deg'][train_ind]                    # w = D^-1 v rides along as the stop column
                err0 = 0.0
                if self.min_iter == 0:                                    # the stop test before the first sweep (ssl.py:667)
                    v = np.zeros(n)
                    v[train_ind] = vval
                    err0 = np.max(np.absolute(v - aux['vinf']))
                aux['sweep'].set_problem_rows(train_ind, Db_rows, w0_rows, err0)
                T, _ = aux['sweep'].run()
                u = _DeviceState(aux['sweep'])
                T, u = self._settle_stop(dev, aux, T, u, train_ind, train_labels)
            else:   # repeated labelled rows, or vertices of degree 0 (D^-1 = inf turns their rows of Db into NaN): the dense expressions, literally
                source, k = _poisson_source(n, train_ind, train_labels)
                Db = aux['D'] * source
                v = np.zeros(n)
                v[train_ind] = 1
                v = v / np.sum(v)
                aux['sweep'].set_problem(Db, v / aux['deg'], aux['deg'], aux['vinf'])
                T, _ = aux['sweep'].run()
                u = _DeviceState(aux['sweep'])
                T, u = self._settle_stop(dev, aux, T, u, train_ind, train_labels)
            self.num_iter = T
            if all_labels is not None and not self.use_cuda:
                # verbose contract of the reference's CPU loop (ssl.py:672-677): one '%d,Accuracy = %.2f' line per
                # sweep.  T is known from the run above; the sweeps are repeated one launch at a time on a sweep
                # without a stop column (the same kernel, the same iterates) and every iterate is read back.
                step = _hip.Sweep(dev, k, min_iter=0, max_iter=0, use_hipgraph=False)
                if Db is None:
                    Db = aux['D'] * _poisson_source(n, train_ind, train_labels)[0]
                try:
                    step.set_state(None, Db)
                    for t in range(1, T + 1):
                        step.iterate(1)
                        self.prob = step.fetch()
                        acc = ssl_accuracy(self.predict(), all_labels, train_ind)
                        print('%d,Accuracy = %.2f' % (t, acc))
                finally:
                    step.close()
                if T > 0:
                    u = self.prob
                elif isinstance(u, _DeviceState):      # no sweep ran: u = 0 (ssl.py:645), not whatever a previous fit left in `prob`
                    u = np.zeros((n, k), dtype=self._dtype())
        elif self.solver == 'spectral':
            raise NotImplementedError("poisson(solver='spectral') needs an eigensolver, which is outside the "
                                      'GPU hot path this package covers (SURVEY.md section 8)')
        else:
            sys.exit('Invalid Poisson solver ' + self.solver)
        return u


    def _settle_stop(self, dev, aux, T, u, train_ind, train_labels):
        """The sweep kernel evaluates the reference's stop test max|v_t - v_inf| > 1/n (ssl.py:667) on a fused column,
        v_t = deg * (P^t D^-1 v_0), while the reference iterates v <- RW v (ssl.py:644, 669): the same numbers up to
        rounding (measured: relative difference <= 1e-13 over whole runs, tests/test_gpu_round2.py).  The two can only
        decide differently when a tested value lies within that rounding of 1/n.  If one lies within STOP_BAND (relative,
        four orders of magnitude wider) the iteration count is derived again with the reference's own recurrence -- RW
        built by the reference's expression, products on the device in csc_matvec's summation order -- and, should it
        differ, exactly that many sweeps are run.  Expected once in ~1e7 fits; every other fit pays one small copy."""
        n = self.graph.num_nodes
        first, vals = aux['sweep'].stop_values()
        self.stop_settled = None
        if not np.any(np.absolute(vals - 1.0 / n) <= STOP_BAND * (1.0 / n)):
            return T, u
        T_exact = self._exact_stop_iteration(train_ind)
        self.stop_settled = (T, T_exact)
        if T_exact == T:
            return T, u
        source, k = _poisson_source(n, train_ind, train_labels)
        v = np.zeros(n)
        v[train_ind] = 1
        v = v / np.sum(v)
        fixed = _hip.Sweep(dev, k, min_iter=T_exact, max_iter=T_exact, use_hipgraph=False)
        try:
            fixed.set_problem(aux['D'] * source, v / aux['deg'], aux['deg'], aux['vinf'])
            fixed.run()
            u = np.array(fixed.fetch())
        finally:
            fixed.close()
        return T_exact, u

    def _exact_stop_iteration(self, train_ind):
        """T of the reference's loop (ssl.py:634-644, 667-670), from its own stop recurrence: RW = W^T D^-1 by the
        reference's expression, v <- RW*v as one device SpMV per step that adds each row's terms in the order scipy's
        csc_matvec does (ascending column of the CSC matrix RW = ascending entry of the sorted CSR), the comparison in
        numpy as written there."""
        n = self.graph.num_nodes
        W = self.graph.weight_matrix
        W = W - sparse.spdiags(W.diagonal(), 0, n, n)
        G = graph_mod.graph(W)
        D = G.degree_matrix(p=-1)
        deg = G.degree_vector()
        RW = sparse.csr_matrix(W.transpose() * D)
        RW.sort_indices()
        rw = _hip.DeviceGraph(RW, dtype=np.float64, device=self.device, keep_order=True)
        try:
            v = np.zeros(n)
            v[train_ind] = 1
            v = v / np.sum(v)
            vinf = deg / np.sum(deg)
            T = 0
            while (T < self.min_iter or np.max(np.absolute(v - vinf)) > 1 / n) and (T < self.max_iter):
                v = rw.spmm_bias(v)
                T = T + 1
        finally:
            rw.close()
        return T

    def _trial_batch_size(self, labels):
        k = max(1, len(np.unique(labels)))
        if self.solver == 'gradient_descent':
            return _gd_batch(k, self._dtype())
        if self.solver != 'conjugate_gradient':
            return 1
        return max(1, min(24, 240 // k))

    def _fit_batch_device(self, trials):
        if self.solver == 'gradient_descent':
            return self._fit_batch_gd(trials)
        return self._fit_batch(trials)

    def _fit_batch_gd(self, trials):
        """Gradient descent (reference ssl.py:631-670) for several training sets as column groups of ONE sweep (glx_sweep_groups):
        trial b owns C columns and its own stop value, runs exactly the T sweeps its own fit would and keeps its iterate from then
        on.  Returns per trial a _DeviceState (the iterate stays in the stacked state until the next solve), or None when the
        trials cannot be stacked (different class counts, repeated labelled rows, vertices of degree 0: the single fit's dense
        branch handles those)."""
        if self.max_iter <= 0 or len(trials) < 2:
            return None
        n = self.graph.num_nodes
        trials = [(np.asarray(ti), np.asarray(tl)) for ti, tl in trials]
        ks = [len(np.unique(tl)) for _, tl in trials]
        k = ks[0]
        if any(kk != k for kk in ks):
            return None
        dev, aux = self._operators()
        if aux['zero_degree']:
            return None
        for ti, _ in trials:
            if not (len(ti) > 0 and len(np.unique(ti)) == len(ti) and ti.min() >= 0 and ti.max() < n):
                return None
        cap = max(len(trials), _gd_batch(k, self._dtype()))
        if cap < 2 or not _gd_fits(k, cap, self._dtype()):
            return None
        key = (k, self.min_iter, self.max_iter, cap)
        if aux.get('groups_key') != key:
            if aux.get('groups') is not None:
                aux['groups'].close()
            aux['groups'] = _hip.SweepGroups(dev, k, cap, min_iter=self.min_iter, max_iter=self.max_iter)
            aux['groups'].set_vectors(aux['deg'], aux['vinf'])
            aux['groups_key'] = key
        groups = aux['groups']
        for b, (ti, tl) in enumerate(trials):
            onehot = utils.labels_to_onehot(tl, k)
            Db_rows = aux['dinv'][ti, None] * (onehot - np.mean(onehot, axis=0))      # as in _fit: rows of D*source
            vval = 1.0 / float(len(ti))
            w0_rows = vval / aux['deg'][ti]
            err0 = 0.0
            if self.min_iter == 0:
                v = np.zeros(n)
                v[ti] = vval
                err0 = np.max(np.absolute(v - aux['vinf']))
            groups.set_problem_rows(b, ti, Db_rows, w0_rows, err0)
        T, _ = groups.run(used=len(trials))
        res = [_DeviceState(groups.view(b)) for b in range(len(trials))]
        settled = []
        for b, (ti, tl) in enumerate(trials):
            # the one case in which the fused stop value and the reference's recurrence could decide differently (_settle_stop):
            # that trial is fitted alone, by the path that settles it
            _, vals = groups.stop_values(b)
            if np.any(np.absolute(vals - 1.0 / n) <= STOP_BAND * (1.0 / n)):
                u = self._fit(ti, tl)
                res[b] = np.array(u.fetch()) if isinstance(u, _DeviceState) else u
                T[b] = self.num_iter
                settled.append(b)
        self.num_iter = [int(t) for t in T]
        self.stop_settled = settled or None
        return res

    def _fit_batch(self, trials):
        """Poisson CG for several training sets at once: the trials' right-hand sides become column
        groups of one multi-RHS solve (glx_cg_groups_masked), each group with the stop test and iteration
        count utils.conjgrad would give it alone (reference: one conjgrad call per trial,
        ssl.py:624-629 under ssl.py:292-396).  solver='gradient_descent': the trials as column groups of one sweep
        (_fit_batch_gd), results read back."""
        if self.solver == 'gradient_descent':
            res = self._fit_batch_gd(trials)
            if res is None:
                return None
            return [np.array(r.fetch()) if isinstance(r, _DeviceState) else r for r in res]
        if self.solver != 'conjugate_gradient':
            return None
        n = self.graph.num_nodes
        sources = [_poisson_source(n, np.asarray(ti), np.asarray(tl)) for ti, tl in trials]
        k = sources[0][1]
        if any(kk != k for _, kk in sources):
            return None
        dev, aux = self._operators()
        D = aux['D']
        B = np.ascontiguousarray(np.hstack([D * src for src, _ in sources]), dtype=self._dtype())
        x, its, _ = dev.cg_groups(B, k, tol=self.tol)
        self.num_iter = [int(i) for i in its]
        return [np.ascontiguousarray(D * x[:, b * k:(b + 1) * k]) for b in range(len(trials))]


class poisson_mbo(ssl):
    def __init__(self, W=None, class_priors=None, solver='conjugate_gradient', use_cuda=False, min_iter=50,
                 max_iter=1000, tol=1e-3, spectral_cutoff=10, Ns=40, mu=1, T=20):
        """PoissonMBO, reference ssl.py:695-839.  class_priors must be provided."""
        super().__init__(W, class_priors)
        self.poisson_model = poisson(W, solver=solver, use_cuda=use_cuda, min_iter=min_iter, max_iter=max_iter,
                                     tol=tol, spectral_cutoff=spectral_cutoff)
        self.Ns = Ns
        self.mu = mu
        self.T = T
        self.use_cuda = use_cuda
        fname = '_poisson_mbo'
        if solver == 'spectral':
            fname += '_N%d' % spectral_cutoff
            self.requries_eig = True
        fname += '_Ns_%d_mu_%.2f_T_%d' % (Ns, mu, T)
        self.accuracy_filename = fname
        self.name = 'Poisson MBO'

    def _fit(self, train_ind, train_labels, all_labels=None):
        Ns, mu, T = self.Ns, self.mu, self.T
        dtype = np.float32 if self.use_cuda else np.float64
        n = self.graph.num_nodes
        W = self.graph.weight_matrix
        source, k = _poisson_source(n, train_ind, train_labels)
        # initialise with Poisson learning (plain argmax: the inner model has no priors)
        labels = self.poisson_model.fit_predict(train_ind, train_labels, all_labels=all_labels)
        # heat operator P = I - dt L and its device image depend on the graph only: kept across fits
        key = (self._graph_key(), dtype, k)
        if self._cache is None or self._cache[0] != key:
            if self._cache is not None:
                self._cache[2].close()
                self._cache[1].close()
            order = getattr(W, '_glx_order', None) if (dtype == np.float64 or n < (1 << 17)) else None   # (as in poisson._operators)
            if order is not None and len(order) != n:
                order = None
            W = W - sparse.spdiags(W.diagonal(), 0, n, n)       # reference ssl.py:789-791
            G = graph_mod.graph(W)
            dt = 1 / np.max(G.degree_vector())                  # reference ssl.py:801
            P = sparse.identity(n) - dt * G.laplacian()         # reference ssl.py:804
            dev = _hip.DeviceGraph(P, dtype=dtype, device=self.device, order=order)     # the search's cell order instead of a pass over the graph
            heat = _hip.Sweep(dev, k, min_iter=0, max_iter=0, use_hipgraph=True)
            self._cache = (key, dev, heat, dt)
        _, dev, heat, dt = self._cache
        if self.class_priors is None:   # the reference fails in volume_label_projection (None arithmetic, ssl.py:199)
            raise TypeError('poisson_mbo needs class_priors for its volume-constrained thresholding')
        # the state never leaves the device between the heat sweeps and the thresholding: the
        # volume-constrained decision runs on the sweep's buffer and writes onehot(labels) back
        # into it (glx_sweep_project_iterate); per outer step only the class weights come back
        rows = np.asarray(train_ind).reshape(-1)
        if len(np.unique(rows)) == len(rows) and labels.min() >= 0 and labels.max() < k:
            # u = onehot(labels) is formed on the device from the labels, Db = mu * dt * source from its labelled rows (every other
            # row of the reference's dense product is a zero): n labels + m rows go up, not two dense (n, k) arrays
            heat.set_state_labels(labels, rows, mu * dt * source[rows])
        else:
            heat.set_state(utils.labels_to_onehot(labels, k), mu * dt * source)    # reference ssl.py:798, 805
        if T > 0:
            heat.iterate(Ns)                                # Ns x `u = P*u + Db`, reference ssl.py:826-827
        for i in range(T):
            w = np.ones((k,)) if type(self.weights) == int else self.weights
            # the thresholding (reference ssl.py:830-832) hands the next outer step's Ns sweeps to the device together with the
            # one-hot state it writes: no host round trip between the two
            labels, w, err, _ = heat.project(self.class_priors, w, max_steps=10000, similarity=self.similarity, to_onehot=True,
                                             want_labels=all_labels is not None, then_iterate=Ns if i + 1 < T else 0)
            self.weights = w
            self.class_priors_error = err
            if all_labels is not None:
                acc = ssl_accuracy(labels, all_labels, train_ind)
                print('%d, Accuracy = %.2f' % (i, acc))
        return heat.fetch().astype(np.float64)              # onehot of the last labels (reference ssl.py:832)


def _neg_columns_times(L, Lcsc, cols, F):
    """`-L[:, cols] * F` (reference ssl.py:1236) without scipy's column fancy-indexing of a CSR
    matrix (7 ms at 60 000 vertices, per training set).  scipy forms the CSR matrix -L[:, cols] --
    row i keeps its selected entries in stored order -- and csr_matvecs adds a row's products one
    after another starting from 0.  With canonical (column-sorted) rows and distinct `cols` that is:
    for every row, the terms (-l_ij) * F[pos(j), :] in ascending j.  The same sums are formed here
    column by column from the CSC image, visiting the selected columns in ascending j (a row occurs
    at most once per column, so each `+=` is one sequential step of that row's sum).  Falls back to
    the literal expression whenever the preconditions do not hold."""
    cols = np.asarray(cols)
    if not (L.has_sorted_indices and L.has_canonical_format and len(np.unique(cols)) == len(cols)):
        return -L[:, cols] * F
    out = np.zeros((L.shape[0], F.shape[1]))
    indptr, indices, data = Lcsc.indptr, Lcsc.indices, Lcsc.data
    for pos in np.argsort(cols, kind='stable'):
        j = cols[pos]
        lo, hi = indptr[j], indptr[j + 1]
        rows = indices[lo:hi]
        out[rows, :] += (-data[lo:hi])[:, None] * F[pos, :][None, :]
    return out


def _neg_columns_times_rows(L, Lcsc, cols, F):
    """The nonzero rows of `-L[:, cols] * F` (see _neg_columns_times) as (row numbers ascending, values): a row's terms
    (-l_ij) * F[pos(j), :] are added in ascending j starting from 0, exactly as there, but only the rows some selected column
    reaches are ever touched.  Returns None when the preconditions of the column-wise form do not hold."""
    cols = np.asarray(cols)
    if not (L.has_sorted_indices and L.has_canonical_format and len(np.unique(cols)) == len(cols)) or len(cols) == 0:
        return None
    indptr, indices, data = Lcsc.indptr, Lcsc.indices, Lcsc.data
    pos = np.argsort(cols, kind='stable')                    # selected columns in ascending j
    cs = cols[pos]
    lo = indptr[cs].astype(np.int64)
    lens = indptr[cs + 1].astype(np.int64) - lo
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int32), np.zeros((0, F.shape[1]))
    starts = np.cumsum(lens) - lens
    idx = np.repeat(lo - starts, lens) + np.arange(total, dtype=np.int64)      # entries of the selected columns, column after column
    rows = indices[idx]
    fpos = np.repeat(pos, lens)
    order = np.argsort(rows, kind='stable')                  # by row; inside a row the ascending-j order survives
    rows_s = rows[order]
    contrib = (-data[idx[order]])[:, None] * F[fpos[order], :]
    first = np.ones(total, dtype=bool)
    first[1:] = rows_s[1:] != rows_s[:-1]
    seg = np.cumsum(first) - 1                               # which output row an entry belongs to
    seg_start = np.flatnonzero(first)
    rank = np.arange(total) - seg_start[seg]                 # its place in that row's sum
    out = np.zeros((len(seg_start), F.shape[1]))
    for r in range(int(rank.max()) + 1):                     # sequential per row: at most one entry per row and pass
        sel = rank == r
        out[seg[sel], :] += contrib[sel, :]
    return rows_s[seg_start].astype(np.int32), out


AUTO_TREE_MAX_ITER = 80
# relative half-width around `tol` inside which a stop decision of the tolerance mode is not trusted to be the reference's: within
# a few dozen iterations the two orders of additions move a residual norm by 1e-13 .. 1e-9 relative.  (Beyond ~90 iterations they
# drift apart by per cent -- conjugate gradients amplify a rounding difference with every iteration: 6 of 863 random systems of
# scripts/auto_margin_probe.py stopped one or two iterations apart, all between 95 and 187 iterations with margins of 0.4 .. 3 % --
# which no band catches at a bearable price; those solves are what AUTO_TREE_MAX_ITER hands back.  profiles/r05_auto_margins.txt)
AUTO_STOP_BAND = 1e-3


def _solve(run, reduce, dev=None):
    """run(mode) -> (x, iterations, err).  reduce='auto': the tolerance mode ('tree'), handed back to the reference-order reductions
    ('exact') when its answer is not one the contract (labels, iteration count, 1e-5) covers -- a solve of more than
    AUTO_TREE_MAX_ITER iterations (every iteration amplifies the difference of the reordered sums: beyond ~90 iterations one solve
    in a hundred stops an iteration apart, profiles/r05_auto_margins.txt), a non-finite iterate (a singular system breaks down with another NaN
    pattern), or ANY comparison of a residual norm with tol -- at every iteration, not only the deciding one: CG's residual norms are not
    monotone -- that hung on less than AUTO_STOP_BAND of tol (`dev`: the operator, asked for the margin of its last tolerance-mode solve).  Well-conditioned systems (config 3: 54 iterations) never go back: they are the 7x faster mode."""
    if reduce != 'auto':
        return run(reduce)
    out = run('tree')
    close = dev is not None and dev.last_stop_margin() < AUTO_STOP_BAND
    if close or int(np.max(out[1])) > AUTO_TREE_MAX_ITER or not np.isfinite(np.asarray(out[0])).all():
        out = run('exact')
    return out


class laplace(ssl):
    def __init__(self, W=None, class_priors=None, X=None, reweighting='none', normalization='combinatorial', tau=0,
                 order=1, mean_shift=False, tol=1e-5, alpha=2, zeta=1e7, r=0.1, reduce='auto'):
        """Laplace learning, reference ssl.py:1106-1261: Dirichlet sub-system solved by a
        Jacobi-scaled multi-RHS conjugate gradient on the GPU.  Reweightings 'poisson' and 'wnll'
        (graph.reweight, reference graph.py:368-466) and 'properly' (needs the features `X`) are supported.

        reduce (not in the reference): 'auto' (default) = the tolerance mode 'tree' -- block-tree reductions, about 7x faster per
        fit at config 3, the same labels and iteration counts, iterates within the north star's 1e-5 of the reference's (this SPD
        system converges to tol=1e-5 either way; include/glx.h GLX_CG_TREE) -- handed back to 'exact' when a solve runs beyond
        ssl.AUTO_TREE_MAX_ITER iterations, produces a non-finite iterate or stops on a margin below ssl.AUTO_STOP_BAND (see
        _solve: the ways the modes were found to part, profiles/r05_auto_margins.txt, tests/test_gpu_auto.py); 'exact' keeps numpy's reduction order: iterates and
        iteration counts bit-identical to the reference."""
        super().__init__(W, class_priors)
        self.reduce = reduce
        self.reweighting = reweighting
        self.normalization = normalization
        self.mean_shift = mean_shift
        self.tol = tol
        self.order = order
        self.X = X
        if type(tau) in [float, int]:
            self.tau = np.ones(self.graph.num_nodes) * tau
        elif type(tau) is np.ndarray:
            self.tau = tau
        fname = '_laplace'
        self.name = 'Laplace Learning'
        if self.reweighting != 'none':
            fname += '_' + self.reweighting
            self.name += ': ' + self.reweighting + ' reweighted'
        if self.normalization != 'combinatorial':
            fname += '_' + self.normalization
            self.name += ' ' + self.normalization
        if self.mean_shift:
            fname += '_meanshift'
            self.name += ' with meanshift'
        if self.order > 1:
            fname += '_order%d' % int(self.order)
            self.name += ' order %d' % int(self.order)
        if np.max(self.tau) > 0:
            fname += '_tau_%.3f' % np.max(self.tau)
            self.name += ' tau=%.3f' % np.max(self.tau)
        self.accuracy_filename = fname
        self.num_iter = None
        self.dtype = np.float64

    def _laplacian(self, G):
        n = G.num_nodes
        L = sparse.spdiags(self.tau, 0, n, n) + G.laplacian(normalization=self.normalization)
        if self.order > 1:                                   # host-side SpGEMM, reference ssl.py:1223-1228
            Lpow = L * L
            for _ in range(2, self.order):
                Lpow = L * Lpow
            L = Lpow
        return L

    def _full_system(self):
        """Without reweighting the operator of every training set is a sub-matrix of ONE matrix:
        A = L[unl, unl] and M = diag((A_ii + 1e-10)^-1/2) (reference ssl.py:1239-1246) are the rows
        and columns of L and of M_full = diag((L_ii + 1e-10)^-1/2) at the unlabelled vertices, so
        M A M is that part of M_full L M_full (same products, same entry order).  Built and
        uploaded once per graph; the solves then hold the labelled rows at zero (glx_cg_groups_masked)."""
        key = (self._graph_key(), self.normalization, np.asarray(self.tau, dtype=np.float64).tobytes(), int(self.order))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1:]
        if self._cache is not None:
            self._cache[3].close()
        n = self.graph.num_nodes
        L = sparse.csr_matrix(self._laplacian(self.graph))
        Mv = 1 / np.sqrt(L.diagonal() + 1e-10)
        M = sparse.spdiags(Mv, 0, n, n).tocsr()
        order = _free_order(self.graph.weight_matrix, n)
        dev = _hip.DeviceGraph(M * L * M, dtype=self.dtype, device=self.device, keep_order=order is None, order=order)
        self._cache = (key, L, Mv, dev)
        return L, Mv, dev

    def _full_csc(self, L):
        if getattr(self, '_csc', None) is None or self._csc[0] is not L:
            self._csc = (L, L.tocsc())
        return self._csc[1]

    def _rhs(self, L, Mv, train_ind, train_labels):
        """F, and M b embedded in an (n, k) array that is zero on the labelled rows (reference ssl.py:1229-1237, 1249)."""
        n = L.shape[0]
        k = len(np.unique(train_labels))
        F = utils.labels_to_onehot(train_labels, k)
        b = _neg_columns_times(L, self._full_csc(L), train_ind, F)   # b = -L[:,train_ind]*F, reference ssl.py:1236
        B = Mv[:, None] * b                                  # row i of M*b is m_i * b_i
        B[train_ind, :] = 0
        return F, np.ascontiguousarray(B, dtype=self.dtype), k

    def _assemble(self, x, Mv, train_ind, F):
        u = Mv[:, None] * x                                  # v = M*v, reference ssl.py:1250
        u[train_ind, :] = F                                  # reference ssl.py:1253-1255
        if self.mean_shift:
            u -= np.mean(u, axis=0)
        return u

    def _fit(self, train_ind, train_labels, all_labels=None):
        if self.reweighting == 'none':
            L, Mv, dev = self._full_system()
            train_ind = np.asarray(train_ind)
            k = len(np.unique(train_labels))
            F = utils.labels_to_onehot(train_labels, k)
            sp = None
            if L.has_sorted_indices and L.has_canonical_format and len(train_ind):
                # b = -L[:,train_ind]*F (ssl.py:1236) lives on the neighbours of the labelled vertices: only those rows go up, the
                # labelled rows themselves left out, M*b (ssl.py:1249) row by row -- one pass of the library's host loop over the
                # selected columns (the numpy form of the same sums, _neg_columns_times_rows, took 0.35 ms of a 3.1 ms fit)
                sp = _hip.host_neg_columns_rows(self._full_csc(L), train_ind, F, row_scale=Mv)
            if sp is not None:
                # `v = M*v` (ssl.py:1250) is applied on the device on the way out
                rows, Mb = sp
                u, its, _ = _solve(lambda mode: dev.cg_groups_rows(rows, Mb, k, masks=[train_ind], out_scale=Mv, tol=self.tol, reduce=mode),
                                   self.reduce, dev)
                self.num_iter = int(its[0])
                u[train_ind, :] = F                              # reference ssl.py:1253-1255
                if self.mean_shift:
                    u -= np.mean(u, axis=0)
                return u
            F, B, k = self._rhs(L, Mv, train_ind, train_labels)
            x, its, _ = _solve(lambda mode: dev.cg_groups(B, k, tol=self.tol, masks=[train_ind], reduce=mode), self.reduce, dev)
            self.num_iter = int(its[0])
            return self._assemble(x, Mv, train_ind, F)
        # reweighted graphs depend on the training set: per-fit sub-matrix, reference ssl.py:1211-1250 line by line
        W = self.graph.reweight(train_ind, method=self.reweighting, normalization=self.normalization, X=self.X)
        G = graph_mod.graph(W)
        n = G.num_nodes
        k = len(np.unique(train_labels))
        L = self._laplacian(G)
        F = utils.labels_to_onehot(train_labels, k)
        idx = np.full((n,), True, dtype=bool)
        idx[train_ind] = False
        b = -L[:, train_ind] * F                             # reference ssl.py:1236-1237
        b = b[idx, :]
        A = L[idx, :]
        A = A[:, idx]
        m = A.shape[0]
        M = A.diagonal()
        M = sparse.spdiags(1 / np.sqrt(M + 1e-10), 0, m, m).tocsr()   # reference ssl.py:1244-1246
        dev = _hip.DeviceGraph(M * A * M, dtype=self.dtype, device=self.device, keep_order=True)
        try:
            rhs = np.ascontiguousarray(M * b, dtype=self.dtype)
            v, it, _ = _solve(lambda mode: dev.cg(rhs, tol=self.tol, reduce=mode), self.reduce, dev)   # reference ssl.py:1249
        finally:
            dev.close()
        self.num_iter = it
        v = M * v
        u = np.zeros((n, k))
        u[idx, :] = v
        u[train_ind, :] = F
        if self.mean_shift:
            u -= np.mean(u, axis=0)
        return u

    def _trial_batch_size(self, labels):
        if self.reweighting != 'none':
            return 1
        k = max(1, len(np.unique(labels)))
        return max(1, min(24, 240 // k))

    def _fit_batch(self, trials):
        """Laplace learning for several training sets at once: one operator, the trials as column
        groups with their own Dirichlet rows, stop tests and iteration counts (glx_cg_groups_masked)."""
        if self.reweighting != 'none':
            return None
        L, Mv, dev = self._full_system()
        parts = [self._rhs(L, Mv, np.asarray(ti), np.asarray(tl)) for ti, tl in trials]
        k = parts[0][2]
        if any(p[2] != k for p in parts):
            return None
        Bs, masks = np.hstack([p[1] for p in parts]), [np.asarray(ti) for ti, _ in trials]
        x, its, _ = _solve(lambda mode: dev.cg_groups(Bs, k, tol=self.tol, masks=masks, reduce=mode), self.reduce, dev)
        self.num_iter = [int(i) for i in its]
        return [self._assemble(np.ascontiguousarray(x[:, j * k:(j + 1) * k]), Mv, np.asarray(trials[j][0]), parts[j][0])
                for j in range(len(trials))]


class randomwalk(ssl):
    def __init__(self, W=None, class_priors=None, alpha=0.95, reduce='auto'):
        """Lazy random walk classification (reference ssl.py:1731-1793): one Jacobi-scaled
        multi-RHS conjugate-gradient solve, on the GPU.  reduce (not in the reference): 'auto' (default) / 'tree' / 'exact', the
        modes of the reductions for this SPD system, see ssl.laplace."""
        super().__init__(W, class_priors)
        self.alpha = alpha
        self.reduce = reduce
        self.accuracy_filename = '_randomwalk'
        self.name = 'Lazy Random Walks'
        self.num_iter = None

    def _operator(self):
        """M L M of reference ssl.py:1779-1786 depends on the graph and alpha only: built and
        uploaded once, shared by every fit on this graph."""
        key = (self._graph_key(), float(self.alpha))
        if self._cache is not None and self._cache[0] == key:
            return self._cache[1], self._cache[2]
        if self._cache is not None:
            self._cache[2].close()
        alpha = self.alpha
        n = self.graph.num_nodes
        W = self.graph.weight_matrix
        W = W - sparse.spdiags(W.diagonal(), 0, n, n)
        G = graph_mod.graph(W)
        L = (1 - alpha) * sparse.identity(n) + alpha * G.laplacian(normalization='normalized')
        m = L.shape[0]
        M = L.diagonal()
        M = sparse.spdiags(1 / np.sqrt(M + 1e-10), 0, m, m).tocsr()
        order = _free_order(self.graph.weight_matrix, n)
        dev = _hip.DeviceGraph(M * L * M, dtype=np.float64, device=self.device, keep_order=order is None, order=order)
        self._cache = (key, M, dev)
        return M, dev

    def _rhs(self, train_ind, train_labels):
        n = self.graph.num_nodes
        k = len(np.unique(train_labels))
        onehot = utils.labels_to_onehot(train_labels, k)
        Y = np.zeros((n, onehot.shape[1]))
        Y[train_ind, :] = onehot
        return Y

    def _fit(self, train_ind, train_labels, all_labels=None):
        M, dev = self._operator()
        rhs = np.ascontiguousarray(M * self._rhs(train_ind, train_labels))
        u, it, _ = _solve(lambda mode: dev.cg(rhs, tol=1e-6, reduce=mode), self.reduce, dev)   # reference ssl.py:1790
        self.num_iter = it
        return M * u

    def _trial_batch_size(self, labels):
        k = max(1, len(np.unique(labels)))
        return max(1, min(24, 240 // k))

    def _fit_batch(self, trials):
        """Several training sets as column groups of one solve (glx_cg_groups_masked), like ssl.poisson."""
        M, dev = self._operator()
        Ys = [M * self._rhs(np.asarray(ti), np.asarray(tl)) for ti, tl in trials]
        k = Ys[0].shape[1]
        if any(Y.shape[1] != k for Y in Ys):
            return None
        Bs = np.hstack(Ys)
        x, its, _ = _solve(lambda mode: dev.cg_groups(Bs, k, tol=1e-6, reduce=mode), self.reduce, dev)
        self.num_iter = [int(i) for i in its]
        return [M * np.ascontiguousarray(x[:, j * k:(j + 1) * k]) for j in range(len(trials))]


def ssl_accuracy(pred_labels, true_labels, train_ind):
    """Accuracy in percent over nodes outside train_ind with a true label >= 0
    (reference ssl.py:1795-1834: `100*np.mean(pred[mask] == true[mask])` over the masked arrays).  The same number from
    counts instead of boolean-indexed copies: np.mean of a 0/1 array is (exact count)/(length) in float64."""
    pred_labels = np.asarray(pred_labels)
    true_labels = np.asarray(true_labels)
    if type(train_ind) != np.ndarray:
        print('Warning: ssl_accuracy requires the indices of the labeled points, not just the number of labels.')
        t = np.zeros(0, dtype=np.int64)
    else:
        t = np.unique(train_ind)
    valid = true_labels >= 0
    hit = (pred_labels == true_labels) & valid
    total = int(np.count_nonzero(valid)) - int(np.count_nonzero(valid[t]))
    hits = int(np.count_nonzero(hit)) - int(np.count_nonzero(hit[t]))
    if total == 0:
        return 100 * np.mean(np.zeros(0, dtype=bool))          # nan (+ numpy's warning), like the reference
    return 100 * (np.float64(hits) / np.float64(total))
