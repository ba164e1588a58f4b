"""ctypes binding of libglx.so (include/glx.h): the only way the package reaches the GPU.

There is NO CPU fallback: if the library is missing or no MI355X is visible the calls
raise GlxError.  The library is loaded lazily (first use), so importing the package
never creates HIP state -- safe for the joblib process pools the reference's
ssl_trials uses (reference graphlearning/ssl.py:390-396).
"""
import os
import ctypes as C
import numpy as np

GLX_F32, GLX_F64 = 0, 1
GLX_CG_NP1D, GLX_CG_TREE, GLX_CG_X0, GLX_CG_BLOCKS, GLX_CG_CHAIN, GLX_CG_EAGER = 1, 2, 4, 8, 16, 32     # flags of glx_cg_solve / glx_cg_groups_masked (include/glx.h)
# how reduce='exact' walks numpy's reduction chains: None = the library's choice (block form from 8192 rows on), 'blocks' / 'chain'
# force one form (same bits; tests and measurements)
CG_EXACT_FORM = None
CG_EXACT_EAGER = False          # True: the exact-mode CG enqueues its iterations launch by launch (GLX_CG_EAGER) instead of replaying captured chunks
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libglx.so')
_lib = None
_default_device = int(os.environ.get('GLX_DEVICE', '0'))


def set_default_device(device):
    """GPU used by every call that is not given an explicit `device` (one process per GPU: set it
    to LOCAL_RANK once).  Initial value: $GLX_DEVICE or 0."""
    global _default_device
    _default_device = int(device)


def default_device():
    return _default_device


def _dev(device):
    if device is None:
        return _default_device
    if hasattr(device, 'index') and not isinstance(device, int):     # a torch.device
        return int(device.index or 0)
    return int(device)


class GlxError(RuntimeError):
    pass


def _dt(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return GLX_F32
    if dtype == np.float64:
        return GLX_F64
    raise GlxError('unsupported dtype %s (float32 or float64)' % dtype)


_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p

# name -> (argtypes); every function returns int status unless listed in _SPECIAL
_SIGNATURES = {
    'glx_version': [],
    'glx_device_count': [C.POINTER(C.c_int)],
    'glx_device_synchronize': [],
    'glx_host_alloc': [C.c_size_t, C.POINTER(_vp)],
    'glx_host_free': [_vp],
    'glx_graph_create': [C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)],
    'glx_graph_create_resident': [C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)],
    'glx_graph_set_row_transform': [_vp, _vp, C.c_int],
    'glx_graph_destroy': [_vp],
    'glx_graph_info': [_vp, _i64p],
    'glx_graph_order': [_vp, _vp],
    'glx_graph_set_order': [_vp, _vp],
    'glx_spmm_bias': [_vp, _vp, _vp, _vp, C.c_int, C.c_int],
    'glx_poisson_sweep': [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)],
    'glx_sweep_create': [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)],
    'glx_sweep_set_problem': [_vp, _vp, _vp, _vp, _vp],
    'glx_sweep_set_vectors': [_vp, _vp, _vp],
    'glx_sweep_set_problem_rows': [_vp, C.c_int64, _vp, _vp, _vp, C.c_double],
    'glx_sweep_run': [_vp, C.POINTER(C.c_int), C.POINTER(C.c_float)],
    'glx_sweep_fetch': [_vp, _vp],
    'glx_sweep_launches': [_vp, _i64p],
    'glx_sweep_stop_values': [_vp, C.c_int64, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    'glx_sweep_destroy': [_vp],
    'glx_sweep_set_state': [_vp, _vp, _vp],
    'glx_sweep_set_state_labels': [_vp, _vp, C.c_int64, _vp, _vp],
    'glx_sweep_iterate': [_vp, C.c_int],
    'glx_sweep_groups_create': [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)],
    'glx_sweep_groups_set_vectors': [_vp, _vp, _vp],
    'glx_sweep_groups_set_problem_rows': [_vp, C.c_int, C.c_int64, _vp, _vp, _vp, C.c_double],
    'glx_sweep_groups_run': [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)],
    'glx_sweep_groups_stop_values': [_vp, C.c_int, C.c_int64, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    'glx_sweep_groups_fetch': [_vp, C.c_int, _vp],
    'glx_sweep_groups_project': [_vp, C.c_int, _vp, _vp, _vp, _f64p, C.POINTER(C.c_int), C.c_int, C.c_int],
    'glx_sweep_groups_launches': [_vp, _i64p],
    'glx_sweep_groups_destroy': [_vp],
    'glx_record_layout': [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)],
    'glx_graph_slots': [_vp, C.c_int, C.c_int, _i64p],
    'glx_bias_flags_dev': [_vp, C.c_int, C.c_int, _vp, _vp, _vp],
    'glx_sweep_step_dev': [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'glx_pack_records_dev': [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp],
    'glx_unpack_records_dev': [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp],
    'glx_rec_dots_dev': [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp],
    'glx_rec_axpy2_dev': [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp],
    'glx_rec_xpby_dev': [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp],
    'glx_dist_unique_id': [_vp],
    'glx_dist_init_rank': [C.c_int, C.c_int, _vp, C.c_int, C.POINTER(_vp)],
    'glx_dist_init': [C.c_int, _vp, C.POINTER(_vp)],
    'glx_dist_comm_info': [_vp, C.POINTER(C.c_int32)],
    'glx_dist_destroy': [_vp],
    'glx_dist_sweep_create': [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int64,
                              C.c_int, C.c_int, C.POINTER(_vp)],
    'glx_dist_sweep_set_problem': [_vp, _vp, _vp, _vp, _vp],
    'glx_poisson_sweep_dist': [_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_float)],
    'glx_dist_sweep_fetch': [_vp, _vp],
    'glx_dist_sweep_stats': [_vp, _i64p],
    'glx_dist_sweep_info': [_vp, _i64p],
    'glx_dist_sweep_time_parts': [_vp, C.c_int, _f32p],
    'glx_dist_sweep_destroy': [_vp],
    'glx_dist_sweep_begin': [_vp],
    'glx_dist_sweep_boundary': [_vp, C.c_int],
    'glx_dist_sweep_get_send': [_vp, _vp],
    'glx_dist_sweep_put_halo': [_vp, _vp, C.c_int],
    'glx_dist_sweep_interior': [_vp, C.c_int, _f64p],
    'glx_cg_multi': [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int), _f64p],
    'glx_cg_solve': [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int64, C.c_int, C.POINTER(C.c_int), _f64p],
    'glx_sweep_project_iterate': [_vp, _vp, _vp, _vp, _f64p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int],
    'glx_lp_iterate': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, C.c_int64, C.c_double, C.c_int64, C.c_int64, C.c_int64,
                       C.POINTER(C.c_int64), C.c_int],
    'glx_affine_iterate': [_vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int64), _f64p],
    'glx_cg_groups_masked': [_vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_int64, C.c_int, C.POINTER(C.c_int), _f64p],
    'glx_cg_groups_rows': [_vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_double, C.c_int64, C.c_int,
                           C.POINTER(C.c_int), _f64p],
    'glx_cg_last_stop_margin': [_vp, _f64p],
    'glx_pool_set_enabled': [C.c_int],
    'glx_pool_set_poison': [C.c_int],
    'glx_debug_set': [C.c_int],
    'glx_debug_counters': [_vp],
    'glx_upload_stats': [_vp],
    'glx_upload_set_mode': [C.c_int],
    'glx_nearest_dist': [_vp, C.c_int64, C.c_int, _vp, C.c_int64, _vp, C.c_int],
    'glx_cg_last_block_stats': [_vp, C.POINTER(C.c_int)],
    'glx_host_fingerprint': [_vp, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64)],
    'glx_knn_search': [_vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)],
    'glx_knn_result_lists': [_vp, _vp, _vp],
    'glx_knn_result_order': [_vp, _vp],
    'glx_knn_result_to_csr': [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int64, _vp, _vp, _vp, C.POINTER(C.c_int64)],
    'glx_knn_result_destroy': [_vp],
    'glx_exp_cr': [_vp, _vp, C.c_int64, C.c_int],
    'glx_argmax_project': [_vp, C.c_int64, C.c_int, _vp, _vp, _vp, _f64p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int],
    'glx_argmax_project_t': [_vp, C.c_int, C.c_int64, C.c_int, _vp, _vp, _vp, _f64p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int],
    'glx_knn_bruteforce': [_vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int],
    'glx_knn_bruteforce_range': [_vp, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, _vp, _vp, C.c_int],
    'glx_knn_cells_range': [_vp, C.c_int64, C.c_int, C.c_int, _vp, C.c_int, C.c_int64, C.c_int64, _vp, _vp, C.c_int],
    'glx_knn_clustered': [_vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int],
    'glx_knn_set_options': [_vp],
    'glx_knn_stats': [_f64p],
    'glx_knn_to_csr': [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp),
                       C.POINTER(_vp), _i64p, C.c_int],
    'glx_knn_rows_to_csr': [_vp, _vp, C.c_int64, C.c_int, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int64, _vp, _vp, _vp,
                            _i64p, C.c_int],
    'glx_host_locality_order': [C.c_int64, _vp, _vp, C.c_int64, _vp],
    'glx_host_permute_rows': [C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'glx_host_row_sums': [C.c_int64, _vp, _vp, _vp],
    'glx_host_neg_columns_rows': [C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int, _vp, C.c_int64, _vp, _vp, _i64p],
    'glx_host_reverse_scale_rows': [C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp],
    'glx_knn_to_csr_into': [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp, _i64p, C.c_int],
}
_SPECIAL = {'glx_last_error': ([], C.c_char_p), 'glx_free': ([_vp], None), 'glx_rec_dots_scratch': ([C.c_int64, C.c_int], C.c_int64)}

EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_SPECIAL))


def load(required=True):
    """Load libglx.so (once).  Raises GlxError when it is absent -- build it with
    `python -m graphlearning_amd._build` or `__graft_entry__.build()`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if required:
            raise GlxError('libglx.so not found at %s: the HIP extension is not built '
                           '(python -m graphlearning_amd._build). There is no CPU fallback.' % LIB_PATH)
        return None
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (args, res) in _SPECIAL.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().glx_last_error()
        raise GlxError('%s failed (%d): %s' % (what or 'libglx call', rc, msg.decode() if msg else '?'))


def device_count():
    n = C.c_int(0)
    check(load().glx_device_count(C.byref(n)), 'glx_device_count')
    return n.value


def require_device():
    try:
        n = device_count()
    except GlxError as e:
        raise GlxError('no usable HIP device: %s' % e)
    if n < 1:
        raise GlxError('no HIP device visible; graphlearning_amd has no CPU fallback')
    return n


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _dense(a, dtype, shape=None, name='array'):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise GlxError('%s has shape %s, expected %s' % (name, a.shape, tuple(shape)))
    return a


class _PinnedBlock:
    """A page-locked host block serving as the memory of one numpy array (its `base`); returned to the pool
    when the array and every view of it are gone."""

    def __init__(self, pool, ptr, nbytes, shape, dtype):
        self._pool, self._ptr, self._nbytes = pool, ptr, nbytes
        self.__array_interface__ = {'data': (ptr, False), 'shape': tuple(shape), 'typestr': np.dtype(dtype).str, 'version': 3}

    def __del__(self):
        try:
            self._pool._release(self._ptr, self._nbytes)
        except Exception:
            pass


class _PinnedPool:
    """Result arrays (prob, labels) are handed out as numpy arrays backed by page-locked memory: the device-to-host
    copy then runs at PCIe speed and a fresh array costs no page faults (a 5.6 MB np.empty + first write is ~1 ms).
    Blocks are recycled by size; at most `keep` idle blocks per size are retained.  Page-locked memory cannot be swapped:
    the pool never holds more than `budget` bytes (live arrays + idle blocks; $GLX_PINNED_MAX_MB, default 4096 MiB) --
    beyond it arrays come from ordinary memory, and idle blocks are released first when a request would not fit."""

    def __init__(self, keep=4):
        import threading
        self.keep = keep
        self.idle = {}
        self.total = 0                                   # bytes of page-locked memory this pool currently owns
        self.budget = int(float(os.environ.get('GLX_PINNED_MAX_MB', '4096')) * (1 << 20))
        self.lock = threading.RLock()                    # graphs are built from several Python threads (and by pinned_reserve's helpers)

    def _trim(self):
        for nbytes, lst in list(self.idle.items()):
            while lst:
                ptr = lst.pop()
                self.total -= nbytes
                if _lib is not None:
                    _lib.glx_host_free(_vp(ptr))

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
        if nbytes < (1 << 16):
            return np.empty(shape, dtype=dtype)
        with self.lock:
            lst = self.idle.get(nbytes)
            ptr = lst.pop() if lst else None
            if ptr is None:
                if self.total + nbytes > self.budget:
                    self._trim()
                if self.total + nbytes > self.budget:
                    return np.empty(shape, dtype=dtype)      # over the budget: ordinary (pageable) memory
                self.total += nbytes                          # (reserved before the allocation: the budget holds across threads)
        if ptr is None:
            p = _vp()
            try:
                check(load().glx_host_alloc(nbytes, C.byref(p)), 'glx_host_alloc')
            except BaseException:
                with self.lock:
                    self.total -= nbytes
                raise
            ptr = p.value
        return np.asarray(_PinnedBlock(self, ptr, nbytes, shape, dtype))

    def _release(self, ptr, nbytes):
        with self.lock:
            lst = self.idle.setdefault(nbytes, [])
            if len(lst) < self.keep:
                lst.append(ptr)
                return
            self.total -= nbytes
        if _lib is not None:
            _lib.glx_host_free(_vp(ptr))


_pinned = _PinnedPool()

# Off switches of the two buffer pools (results are identical either way: tests/test_gpu_switches.py; the randomised soak runs with
# each off as an ablation -- a result that moves with a switch names the subsystem):
#   PINNED_RESULTS = False   result arrays come from np.empty (ordinary memory), no page-locked block is recycled
#   pool_set_enabled(False)  device work buffers and work sets come straight from / go straight back to the HIP runtime
PINNED_RESULTS = True


def pool_set_enabled(enabled):
    check(load().glx_pool_set_enabled(1 if enabled else 0), 'glx_pool_set_enabled')


def debug_set(flags):
    check(load().glx_debug_set(int(flags)), 'glx_debug_set')


def debug_counters():
    out = (C.c_uint64 * 4)()
    check(load().glx_debug_counters(out), 'glx_debug_counters')
    return dict(uploads_checked=int(out[0]), engine_readback_differs=int(out[1]), kernel_readback_differs=int(out[2]), bytes_differing=int(out[3]))


def upload_set_mode(mode):
    """0 staged + checked (default), 1 staged, 2 straight from the caller's pageable memory (rounds 1-5)."""
    check(load().glx_upload_set_mode(int(mode)), 'glx_upload_set_mode')


def upload_stats():
    """The checked uploads of this process: how many were checked, how many arrived with a wrong sum, how many of those were repaired by a
    repeat, how many given up (glx_upload_stats)."""
    out = (C.c_uint64 * 4)()
    check(load().glx_upload_stats(out), 'glx_upload_stats')
    return dict(checked=int(out[0]), wrong_sums=int(out[1]), repaired=int(out[2]), given_up=int(out[3]))


def pool_set_poison(byte):
    """Debugging aid: work buffers are filled with `byte` (0 .. 255) when handed out; None / -1 switches it off."""
    check(load().glx_pool_set_poison(-1 if byte is None else int(byte)), 'glx_pool_set_poison')


def pinned_empty(shape, dtype):
    if not PINNED_RESULTS:
        return np.empty(shape, dtype=np.dtype(dtype))
    return _pinned.empty(shape, dtype)


def pinned_reserve(specs):
    """Make sure the pool holds an idle block for every (shape, dtype) of `specs`, allocating the missing ones on a helper thread
    (page-locking 19 MB of fresh memory takes 3.4 ms: weightmatrix.knn starts it beside its search, whose result arrays these
    are).  Returns the thread to join, or None when nothing had to be allocated (the steady state)."""
    need = []
    for shape, dtype in specs if PINNED_RESULTS else ():
        nbytes = max(int(np.prod(shape)) * np.dtype(dtype).itemsize, 1)
        with _pinned.lock:
            if nbytes >= (1 << 16) and not _pinned.idle.get(nbytes) and _pinned.total + nbytes <= _pinned.budget:
                _pinned.total += nbytes              # reserved now, handed to the idle list by the helper
                need.append(nbytes)
    if not need:
        return None
    lib = load()

    def work(nbytes):
        p = _vp()
        ok = lib.glx_host_alloc(nbytes, C.byref(p)) == 0
        with _pinned.lock:
            if ok:
                _pinned.idle.setdefault(nbytes, []).append(p.value)
            else:
                _pinned.total -= nbytes
    import threading
    ths = [threading.Thread(target=work, args=(nb,), daemon=True) for nb in sorted(need, reverse=True)]   # (one per block: page-locking runs in parallel)
    for th in ths:
        th.start()

    class _Join:
        def join(self):
            for th in ths:
                th.join()
    return _Join()


class DeviceGraph:
    """A sparse operator resident in HBM (glx_graph).  `A` is any scipy sparse matrix;
    the CSR entry order is preserved (see include/glx.h)."""

    def __init__(self, A, dtype=np.float64, device=None, shape=None, keep_order=False, order=None):
        from scipy import sparse
        A = sparse.csr_matrix(A)
        self.dtype = np.dtype(dtype)
        self.shape = A.shape if shape is None else shape
        self.nnz = int(A.nnz)
        self.device = device = _dev(device)
        rowptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        col = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data, dtype=np.float64)
        self._h = _vp()
        lib = load()
        check(lib.glx_graph_create(self.shape[0], self.shape[1], self.nnz, _ptr(rowptr), _ptr(col), _ptr(val),
                                   _dt(self.dtype), device, C.byref(self._h)), 'glx_graph_create')
        self._set_order(order, keep_order)

    def _set_order(self, order, keep_order):
        lib = load()
        if order is not None:       # the caller's locality order (perm[new] = old) instead of the library's pass over the graph
            perm = np.ascontiguousarray(order, dtype=np.int32)
            if perm.shape != (self.shape[0],):
                raise GlxError('order must be a permutation of the %d rows' % self.shape[0])
            check(lib.glx_graph_set_order(self._h, _ptr(perm)), 'glx_graph_set_order')
        elif keep_order:
            check(lib.glx_graph_set_order(self._h, None), 'glx_graph_set_order')

    @classmethod
    def resident(cls, A, dtype=np.float64, device=None, keep_order=False, order=None, want_row_sums=False):
        """The operator of CSR matrix `A` with the arrays kept on the device only (glx_graph_create_resident); with
        want_row_sums also `A * ones` (scipy's csr_matvec order) computed there: returns (graph, row_sums or None).
        `set_row_transform` then turns the rows into what the operator really is (P = D^-1 W^T: reversed rows times 1 / degree)."""
        self = cls.__new__(cls)
        self.dtype = np.dtype(dtype)
        self.shape = A.shape
        self.nnz = int(A.nnz)
        self.device = device = _dev(device)
        rowptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        col = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data, dtype=np.float64)
        sums = np.empty(self.shape[0], dtype=np.float64) if want_row_sums else None
        self._h = _vp()
        check(load().glx_graph_create_resident(self.shape[0], self.shape[1], self.nnz, _ptr(rowptr), _ptr(col), _ptr(val),
                                               _dt(self.dtype), device, _ptr(sums), C.byref(self._h)), 'glx_graph_create_resident')
        self._set_order(order, keep_order)
        return self, sums

    def set_row_transform(self, row_scale=None, reverse_rows=False):
        sc = None if row_scale is None else _dense(row_scale, np.float64, (self.shape[0],), 'row_scale')
        check(load().glx_graph_set_row_transform(self._h, _ptr(sc), 1 if reverse_rows else 0), 'glx_graph_set_row_transform')

    def info(self):
        out = (C.c_int64 * 8)()
        check(load().glx_graph_info(self._h, out), 'glx_graph_info')
        keys = ['n_rows', 'n_cols', 'nnz', 'stored', 'slices', 'rows_per_slice', 'max_row', 'renumbered']
        return dict(zip(keys, list(out)[:8]))

    def order(self):
        """perm[new] = caller's row: the vertex order the library keeps its records in."""
        perm = np.empty(self.shape[0], dtype=np.int32)
        check(load().glx_graph_order(self._h, _ptr(perm)), 'glx_graph_order')
        return perm

    def spmm_bias(self, u, Db=None, iters=1):
        """`iters` applications of u <- Db + A u (host arrays in, host array out)."""
        u = np.ascontiguousarray(u, dtype=self.dtype)
        if u.ndim == 1:
            return self.spmm_bias(u[:, None], None if Db is None else np.asarray(Db)[:, None], iters)[:, 0]
        if u.shape[0] != self.shape[1]:
            raise GlxError('operand has %d rows, operator has %d columns' % (u.shape[0], self.shape[1]))
        Cc = u.shape[1]
        Dbc = None if Db is None else _dense(Db, self.dtype, (self.shape[0], Cc), 'Db')
        out = np.empty((self.shape[0], Cc), dtype=self.dtype)
        check(load().glx_spmm_bias(self._h, _ptr(Dbc), _ptr(u), _ptr(out), Cc, iters), 'glx_spmm_bias')
        return out

    def poisson_sweep(self, Db, w0, deg, vinf, min_iter=50, max_iter=1000):
        Db = np.ascontiguousarray(Db, dtype=self.dtype)
        n, Cc = Db.shape
        w0 = _dense(w0, np.float64, (n,), 'w0')
        deg = _dense(deg, np.float64, (n,), 'deg')
        vinf = _dense(vinf, np.float64, (n,), 'vinf')
        out = np.empty((n, Cc), dtype=self.dtype)
        T = C.c_int(0)
        check(load().glx_poisson_sweep(self._h, _ptr(Db), _ptr(w0), _ptr(deg), _ptr(vinf), Cc, min_iter, max_iter,
                                       _ptr(out), C.byref(T)), 'glx_poisson_sweep')
        return out, T.value

    def cg(self, B, tol=1e-10, max_iter=100000, x0=None, reduce='exact'):
        """utils.conjgrad on device: returns (X, iterations, err).  x0: initial iterate (B must then
        be the residual b - A@x0, utils.py:510-514).  reduce='tree': tolerance mode (GLX_CG_TREE)."""
        B = np.ascontiguousarray(B, dtype=self.dtype)
        squeeze = B.ndim == 1
        if squeeze:
            B = B[:, None]
        # (a single column, 1-D or (n,1): numpy's axis-0 reduction of it is one contiguous run, summed pairwise)
        flags = (GLX_CG_NP1D if B.shape[1] == 1 else 0) | _reduce_flag(reduce)
        if x0 is None:
            X = np.empty_like(B)
        else:
            X = np.array(np.asarray(x0).reshape(B.shape), dtype=self.dtype, order='C', copy=True)
            flags |= GLX_CG_X0
        it = C.c_int(0)
        err = C.c_double(0)
        check(load().glx_cg_solve(self._h, _ptr(B), _ptr(X), B.shape[1], float(tol), int(max_iter), flags,
                                  C.byref(it), C.byref(err)), 'glx_cg_solve')
        return (X[:, 0] if squeeze else X), it.value, err.value

    def affine_iterate(self, u0, b=None, tol=1e-10, max_iter=1000000):
        """u <- A u + b from u0 until max|u_new - u_old| <= tol (graph.page_rank's power iteration):
        returns (u, sweeps, err)."""
        u0 = np.ascontiguousarray(u0, dtype=self.dtype)
        squeeze = u0.ndim == 1
        if squeeze:
            u0 = u0[:, None]
        if b is not None:
            b = np.ascontiguousarray(b, dtype=self.dtype).reshape(u0.shape)
        out = np.empty_like(u0)
        it = C.c_int64(0)
        err = C.c_double(0)
        check(load().glx_affine_iterate(self._h, _ptr(b) if b is not None else None, _ptr(u0), _ptr(out), u0.shape[1], float(tol),
                                        int(max_iter), C.byref(it), C.byref(err)), 'glx_affine_iterate')
        return (out[:, 0] if squeeze else out), it.value, err.value

    def cg_groups(self, B, group_cols, tol=1e-10, max_iter=100000, masks=None, reduce='exact'):
        """Independent systems side by side (columns in groups of `group_cols`), each with its own
        stop test: returns (X, iterations per group, err per group).  masks: per group an array of
        Dirichlet rows (x held at zero there; B is zeroed on them) -- the sub-matrix solve of
        ssl.laplace on the full operator."""
        B = np.ascontiguousarray(B, dtype=self.dtype)
        if masks is not None:
            B = B.copy()                      # zeroed on the Dirichlet rows below; the caller's array stays as it is
        flags = _reduce_flag(reduce)
        ng = B.shape[1] // group_cols
        X = np.empty_like(B)
        its = np.zeros(ng, dtype=np.int32)
        errs = np.zeros(ng, dtype=np.float64)
        if masks is None:
            check(load().glx_cg_groups_masked(self._h, _ptr(B), _ptr(X), B.shape[1], int(group_cols), None, None, float(tol), int(max_iter),
                                              flags, its.ctypes.data_as(C.POINTER(C.c_int)), errs.ctypes.data_as(_f64p)), 'glx_cg_groups_masked')
            return X, its, errs
        if len(masks) != ng:
            raise GlxError('cg_groups: %d masks for %d systems' % (len(masks), ng))
        rows = [np.ascontiguousarray(m, dtype=np.int32).ravel() for m in masks]
        ptr = np.zeros(ng + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([len(r) for r in rows])
        allrows = np.ascontiguousarray(np.concatenate(rows) if rows else np.zeros(0, np.int32), dtype=np.int32)
        for g, r in enumerate(rows):
            B[r, g * group_cols:(g + 1) * group_cols] = 0
        check(load().glx_cg_groups_masked(self._h, _ptr(B), _ptr(X), B.shape[1], int(group_cols), _ptr(allrows), _ptr(ptr),
                                          float(tol), int(max_iter), flags, its.ctypes.data_as(C.POINTER(C.c_int)),
                                          errs.ctypes.data_as(_f64p)), 'glx_cg_groups_masked')
        return X, its, errs

    def cg_groups_rows(self, rows, vals, group_cols, masks, out_scale=None, tol=1e-10, max_iter=100000, reduce='exact'):
        """cg_groups for a right-hand side given by its nonzero rows (`rows` distinct, `vals` (len(rows), C); every other row
        zero -- and zero on the Dirichlet rows, which the caller guarantees), the result optionally scaled row by row on the
        device (glx_cg_groups_rows).  Returns (X, iterations per group, err per group); X is page-locked memory."""
        rows = np.ascontiguousarray(rows, dtype=np.int32).ravel()
        vals = np.ascontiguousarray(vals, dtype=self.dtype)
        Cc = vals.shape[1]
        ng = Cc // group_cols
        if len(masks) != ng:
            raise GlxError('cg_groups_rows: %d masks for %d systems' % (len(masks), ng))
        mrows = [np.ascontiguousarray(m, dtype=np.int32).ravel() for m in masks]
        ptr = np.zeros(ng + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([len(r) for r in mrows])
        allrows = np.ascontiguousarray(np.concatenate(mrows) if mrows else np.zeros(0, np.int32), dtype=np.int32)
        scale = None if out_scale is None else _dense(out_scale, np.float64, (self.shape[0],), 'out_scale')
        X = pinned_empty((self.shape[0], Cc), self.dtype)
        its = np.zeros(ng, dtype=np.int32)
        errs = np.zeros(ng, dtype=np.float64)
        check(load().glx_cg_groups_rows(self._h, len(rows), _ptr(rows), _ptr(vals), _ptr(scale), _ptr(X), Cc, int(group_cols),
                                        _ptr(allrows), _ptr(ptr), float(tol), int(max_iter), _reduce_flag(reduce),
                                        its.ctypes.data_as(C.POINTER(C.c_int)), errs.ctypes.data_as(_f64p)), 'glx_cg_groups_rows')
        return X, its, errs

    def last_block_stats(self):
        """(plain, by record, row by row) block counts of the last reference-order solve's reduction chains; (-1, -1, -1): chain form.
        `last_block_forms()`: which kinds of reduction ended the solve in block form (bit 0: p.Ap, bit 1: r.r)."""
        return self._block_stats()[:3]

    def last_block_forms(self):
        return self._block_stats()[3]

    def _block_stats(self):
        out = (C.c_int * 4)()
        check(load().glx_cg_last_block_stats(self._h, out), 'glx_cg_last_block_stats')
        return tuple(out)

    def last_stop_margin(self):
        """How close the stop decisions of the last tolerance-mode solve on this operator came to going the other way (relative to
        tol; inf: no such solve)."""
        m = C.c_double(0)
        check(load().glx_cg_last_stop_margin(self._h, C.byref(m)), 'glx_cg_last_stop_margin')
        return m.value

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib = load(required=False)
            if lib is not None:
                lib.glx_graph_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_fingerprint(arrays):
    """128-bit content fingerprint of a list of contiguous numpy arrays (glx_host_fingerprint: threaded, in the library), or
    None when the library is not there."""
    lib = load(required=False)
    if lib is None:
        return None
    out = (C.c_uint64 * 2)()
    acc = 0
    for a in arrays:
        check(lib.glx_host_fingerprint(_ptr(a), a.nbytes, acc & 0xffffffffffffffff, out), 'glx_host_fingerprint')
        acc = out[0] ^ ((out[1] * 0x9e3779b97f4a7c15) & 0xffffffffffffffff)
    return (int(out[0]), int(out[1]))


def _reduce_flag(reduce):
    if reduce == 'exact':
        if CG_EXACT_FORM not in (None, 'blocks', 'chain'):
            raise GlxError("CG_EXACT_FORM must be None, 'blocks' or 'chain', got %r" % (CG_EXACT_FORM,))
        return {None: 0, 'blocks': GLX_CG_BLOCKS, 'chain': GLX_CG_CHAIN}[CG_EXACT_FORM] | (GLX_CG_EAGER if CG_EXACT_EAGER else 0)
    if reduce == 'tree':
        return GLX_CG_TREE
    raise GlxError("reduce must be 'exact' (reference-order reductions) or 'tree' (tolerance mode), got %r" % (reduce,))


class Sweep:
    """Prepared Poisson / heat sweep (glx_sweep): device-resident state, repeated runs."""

    def __init__(self, graph, Cc, min_iter=50, max_iter=1000, use_hipgraph=True):
        self.graph = graph
        self.C = Cc
        self._h = _vp()
        check(load().glx_sweep_create(graph._h, Cc, min_iter, max_iter, 1 if use_hipgraph else 0, C.byref(self._h)),
              'glx_sweep_create')

    def set_problem(self, Db, w0, deg, vinf):
        n = self.graph.shape[0]
        Db = _dense(Db, self.graph.dtype, (n, self.C), 'Db')
        check(load().glx_sweep_set_problem(self._h, _ptr(Db), _ptr(_dense(w0, np.float64, (n,))),
                                           _ptr(_dense(deg, np.float64, (n,))), _ptr(_dense(vinf, np.float64, (n,)))),
              'glx_sweep_set_problem')

    def set_vectors(self, deg, vinf):
        """The graph's own vectors of the stop test (ssl.py:642-643), uploaded once per prepared sweep."""
        n = self.graph.shape[0]
        check(load().glx_sweep_set_vectors(self._h, _ptr(_dense(deg, np.float64, (n,))), _ptr(_dense(vinf, np.float64, (n,)))),
              'glx_sweep_set_vectors')

    def set_problem_rows(self, rows, Db_rows, w0_rows, err0):
        """A new training set: the labelled rows, their rows of Db = D^-1 b and of w0 = v0/deg, and
        err0 = max|v0 - vinf| (ssl.py:620-622, 636, 639-641, 667)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64).ravel()
        m = len(rows)
        Db_rows = _dense(Db_rows, self.graph.dtype, (m, self.C), 'Db_rows')
        w0_rows = _dense(w0_rows, np.float64, (m,), 'w0_rows')
        check(load().glx_sweep_set_problem_rows(self._h, m, _ptr(rows), _ptr(Db_rows), _ptr(w0_rows), float(err0)),
              'glx_sweep_set_problem_rows')

    def run(self):
        T = C.c_int(0)
        ms = C.c_float(0)
        check(load().glx_sweep_run(self._h, C.byref(T), C.byref(ms)), 'glx_sweep_run')
        self.generation = getattr(self, 'generation', 0) + 1
        return T.value, ms.value

    def stop_values(self):
        """(first, values): the stop values max|v_t - v_inf| the last run() compared with 1/n, t = first, first+1, ..."""
        first, count = C.c_int(0), C.c_int(0)
        check(load().glx_sweep_stop_values(self._h, 0, None, C.byref(first), C.byref(count)), 'glx_sweep_stop_values')
        vals = np.empty(count.value, dtype=np.float64)
        check(load().glx_sweep_stop_values(self._h, len(vals), _ptr(vals), C.byref(first), C.byref(count)), 'glx_sweep_stop_values')
        return first.value, vals

    def set_state(self, u0, Db=None):
        n = self.graph.shape[0]
        u0 = None if u0 is None else _dense(u0, self.graph.dtype, (n, self.C), 'u0')
        Db = None if Db is None else _dense(Db, self.graph.dtype, (n, self.C), 'Db')
        check(load().glx_sweep_set_state(self._h, _ptr(u0), _ptr(Db)), 'glx_sweep_set_state')
        self.generation = getattr(self, 'generation', 0) + 1

    def set_state_labels(self, labels, rows, Db_rows):
        """u = onehot(labels), bias = its m nonzero rows (distinct `rows`): n labels and m rows go up instead of two dense arrays."""
        n = self.graph.shape[0]
        labels = _dense(labels, np.int64, (n,), 'labels')
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        Db_rows = _dense(Db_rows, self.graph.dtype, (len(rows), self.C), 'Db_rows')
        check(load().glx_sweep_set_state_labels(self._h, _ptr(labels), len(rows), _ptr(rows), _ptr(Db_rows)), 'glx_sweep_set_state_labels')
        self.generation = getattr(self, 'generation', 0) + 1

    def iterate(self, iters):
        check(load().glx_sweep_iterate(self._h, int(iters)), 'glx_sweep_iterate')
        self.generation = getattr(self, 'generation', 0) + 1

    def fetch(self):
        out = pinned_empty((self.graph.shape[0], self.C), self.graph.dtype)
        check(load().glx_sweep_fetch(self._h, _ptr(out)), 'glx_sweep_fetch')
        return out

    def project(self, priors=None, weights=None, max_steps=0, similarity=True, to_onehot=False, want_labels=True, then_iterate=0):
        """ssl.predict / ssl.volume_label_projection on the device-resident state; with to_onehot the
        state becomes onehot(labels), and then_iterate sweeps are enqueued behind it at once (not awaited).
        Returns (labels int64 or None, weights, err, steps)."""
        n = self.graph.shape[0]
        w = np.ones(self.C) if weights is None else np.array(weights, dtype=np.float64).reshape(self.C).copy()
        pri = np.zeros(self.C) if priors is None else _dense(priors, np.float64, (self.C,), 'priors')
        labels = pinned_empty((n,), np.int64) if want_labels else None
        err = C.c_double(0)
        steps = C.c_int(0)
        check(load().glx_sweep_project_iterate(self._h, _ptr(pri), _ptr(w), _ptr(labels) if want_labels else None, C.byref(err),
                                               C.byref(steps), int(max_steps), 1 if similarity else 0, 1 if to_onehot else 0,
                                               int(then_iterate)), 'glx_sweep_project_iterate')
        if to_onehot or then_iterate:
            self.generation = getattr(self, 'generation', 0) + 1
        return labels, w, err.value, steps.value

    def launches(self):
        n = C.c_int64(0)
        check(load().glx_sweep_launches(self._h, C.byref(n)), 'glx_sweep_launches')
        return n.value

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib = load(required=False)
            if lib is not None:
                lib.glx_sweep_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SweepGroups:
    """Several training sets as column groups of ONE prepared Poisson sweep (glx_sweep_groups): trial b owns columns
    b C .. b C + C - 1, its own stop value and its own stop test; every trial's iterate and T are those of its own fit."""

    def __init__(self, graph, Cc, B, min_iter=50, max_iter=1000):
        self.graph = graph
        self.C = Cc
        self.B = B
        self.generation = 0
        self._h = _vp()
        check(load().glx_sweep_groups_create(graph._h, Cc, B, min_iter, max_iter, C.byref(self._h)), 'glx_sweep_groups_create')

    def set_vectors(self, deg, vinf):
        n = self.graph.shape[0]
        check(load().glx_sweep_groups_set_vectors(self._h, _ptr(_dense(deg, np.float64, (n,))), _ptr(_dense(vinf, np.float64, (n,)))),
              'glx_sweep_groups_set_vectors')

    def set_problem_rows(self, b, rows, Db_rows, w0_rows, err0):
        rows = np.ascontiguousarray(rows, dtype=np.int64).ravel()
        m = len(rows)
        Db_rows = _dense(Db_rows, self.graph.dtype, (m, self.C), 'Db_rows')
        w0_rows = _dense(w0_rows, np.float64, (m,), 'w0_rows')
        check(load().glx_sweep_groups_set_problem_rows(self._h, int(b), m, _ptr(rows), _ptr(Db_rows), _ptr(w0_rows), float(err0)),
              'glx_sweep_groups_set_problem_rows')

    def run(self, used=None):
        """Returns (T per group in use, device ms)."""
        used = self.B if used is None else int(used)
        T = (C.c_int * self.B)()
        ms = C.c_float(0)
        check(load().glx_sweep_groups_run(self._h, used, T, C.byref(ms)), 'glx_sweep_groups_run')
        self.generation += 1
        return [int(T[b]) for b in range(used)], ms.value

    def stop_values(self, b):
        first, count = C.c_int(0), C.c_int(0)
        check(load().glx_sweep_groups_stop_values(self._h, int(b), 0, None, C.byref(first), C.byref(count)), 'glx_sweep_groups_stop_values')
        vals = np.empty(count.value, dtype=np.float64)
        check(load().glx_sweep_groups_stop_values(self._h, int(b), len(vals), _ptr(vals), C.byref(first), C.byref(count)),
              'glx_sweep_groups_stop_values')
        return first.value, vals

    def fetch(self, b):
        out = pinned_empty((self.graph.shape[0], self.C), self.graph.dtype)
        check(load().glx_sweep_groups_fetch(self._h, int(b), _ptr(out)), 'glx_sweep_groups_fetch')
        return out

    def project(self, b, priors=None, weights=None, max_steps=0, similarity=True, want_labels=True):
        n = self.graph.shape[0]
        w = np.ones(self.C) if weights is None else np.array(weights, dtype=np.float64).reshape(self.C).copy()
        pri = np.zeros(self.C) if priors is None else _dense(priors, np.float64, (self.C,), 'priors')
        labels = pinned_empty((n,), np.int64) if want_labels else None
        err = C.c_double(0)
        steps = C.c_int(0)
        check(load().glx_sweep_groups_project(self._h, int(b), _ptr(pri), _ptr(w), _ptr(labels) if want_labels else None, C.byref(err),
                                              C.byref(steps), int(max_steps), 1 if similarity else 0), 'glx_sweep_groups_project')
        return labels, w, err.value, steps.value

    def launches(self):
        n = C.c_int64(0)
        check(load().glx_sweep_groups_launches(self._h, C.byref(n)), 'glx_sweep_groups_launches')
        return n.value

    def view(self, b):
        return GroupView(self, b)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib = load(required=False)
            if lib is not None:
                lib.glx_sweep_groups_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupView:
    """Group b of a SweepGroups with the face of a Sweep (C, fetch, project, generation, _h): what ssl._DeviceState needs to leave a
    trial's result on the device until somebody looks at it."""

    def __init__(self, groups, b):
        self.groups, self.b, self.C = groups, int(b), groups.C

    @property
    def _h(self):
        return self.groups._h

    @property
    def generation(self):
        return self.groups.generation

    def fetch(self):
        return self.groups.fetch(self.b)

    def project(self, priors=None, weights=None, max_steps=0, similarity=True, to_onehot=False, want_labels=True, then_iterate=0):
        if to_onehot or then_iterate:
            raise GlxError('a stacked trial has no heat loop of its own')
        return self.groups.project(self.b, priors, weights, max_steps=max_steps, similarity=similarity, want_labels=want_labels)


# Communicators and distributed sweeps that are still open when the interpreter shuts down are ABANDONED, not destroyed:
# tearing an RCCL communicator (or the streams and captured graphs that carry its kernels) down from a garbage-collection
# pass during interpreter finalisation was seen to hang the process (a failed test that never reached close(), round 3);
# the operating system reclaims everything at exit anyway.  close() during normal operation destroys as before.
import atexit
import weakref
_live_dist_objects = weakref.WeakSet()


def _abandon_dist_objects():
    for obj in list(_live_dist_objects):
        try:
            obj._h = _vp()
        except Exception:
            pass


atexit.register(_abandon_dist_objects)


class Comm:
    """One rank's libglx-owned RCCL communicator (glx_comm).  `uid`: the 128 bytes of Comm.unique_id() made on rank 0
    and shipped to every rank; uid=None with nranks == 1 gives a transport-free single rank."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(load().glx_dist_unique_id(buf), 'glx_dist_unique_id')
        return buf.raw

    def __init__(self, nranks=1, rank=0, uid=None, device=None):
        self.nranks, self.rank, self.device = int(nranks), int(rank), _dev(device)
        self._h = _vp()
        idbuf = None if uid is None else C.create_string_buffer(bytes(uid), 128)
        check(load().glx_dist_init_rank(self.nranks, self.rank, idbuf, self.device, C.byref(self._h)), 'glx_dist_init_rank')
        _live_dist_objects.add(self)

    def info(self):
        info = (C.c_int32 * 4)()
        check(load().glx_dist_comm_info(self._h, info), 'glx_dist_comm_info')
        return dict(rank=int(info[0]), nranks=int(info[1]), device=int(info[2]), rccl=bool(info[3]))

    def has_transport(self):
        return self.info()['rccl']

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib = load(required=False)
            if lib is not None:
                lib.glx_dist_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# glx_dist_sweep_create's form flags (include/glx.h): what the library picks by itself, and the forms tests force
DIST_FORMS = {'auto': 0, 'split': 2, 'split_pack': 2 | 8, 'split_inline': 2 | 16, 'fused': 4, 'selftest': 128, 'eager': 64, 'captured': 32}
DIST_GATHER = 256        # GLX_DIST_FORM_GATHER: the exchange is one in-place all-gather of the ranks' blocks (dist.GatherPlan)


class DistSweep:
    """One rank's share of the vertex-partitioned Poisson sweep (glx_dist_sweep): local operator (boundary rows first,
    columns [owned | halo]), exchange lists, device state; every sweep and every collective is enqueued by libglx."""

    def __init__(self, comm, P_local, n_boundary, send_counts, send_idx, recv_counts, n_global, Cc, dtype=np.float64,
                 force_exchange=False, use_hipgraph=True, form='auto', gather_cap=None):
        """gather_cap (dist.GatherPlan): the all-gather form -- P_local's columns are numbered owner * cap + (row within the owner's
        block), shape (n_own, world * cap); the exchange lists are not used."""
        from scipy import sparse
        A = sparse.csr_matrix(P_local)
        self.comm = comm
        self.n_own, n_loc = A.shape
        self.gather_cap = None if gather_cap is None else int(gather_cap)
        self.n_halo = n_loc - (self.n_own if gather_cap is None else self.gather_cap)
        self.C = int(Cc)
        self.dtype = np.dtype(dtype)
        self.lay = record_layout(Cc, dtype, True)
        rowptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        col = np.ascontiguousarray(A.indices, dtype=np.int32)
        val = np.ascontiguousarray(A.data, dtype=np.float64)
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rcnt = np.ascontiguousarray(recv_counts, dtype=np.int64)
        si = np.ascontiguousarray(send_idx, dtype=np.int32)
        self.n_send = int(sc.sum()) if gather_cap is None else self.gather_cap
        self._h = _vp()
        flags = (1 if use_hipgraph else 0) | DIST_FORMS[form] | (DIST_GATHER if gather_cap is not None else 0)
        check(load().glx_dist_sweep_create(comm._h, self.n_own, self.n_halo, int(n_boundary), _ptr(rowptr), _ptr(col), _ptr(val),
                                           _dt(self.dtype), self.C, _ptr(sc), _ptr(si), _ptr(rcnt), int(n_global),
                                           1 if force_exchange else 0, flags, C.byref(self._h)),
              'glx_dist_sweep_create')
        _live_dist_objects.add(self)

    def set_problem(self, Db_own, w0_own, deg_own, vinf_own):
        n = self.n_own
        Db = None if Db_own is None else _dense(Db_own, self.dtype, (n, self.C), 'Db_own')
        check(load().glx_dist_sweep_set_problem(self._h, _ptr(Db), _ptr(_dense(w0_own, np.float64, (n,))),
                                                _ptr(_dense(deg_own, np.float64, (n,))), _ptr(_dense(vinf_own, np.float64, (n,)))),
              'glx_dist_sweep_set_problem')

    def run(self, min_iter=50, max_iter=1000, check_every=8, err0=0.0):
        T = C.c_int(0)
        ms = C.c_float(0)
        check(load().glx_poisson_sweep_dist(self._h, int(min_iter), int(max_iter), int(check_every), float(err0), C.byref(T),
                                            C.byref(ms)), 'glx_poisson_sweep_dist')
        return T.value, ms.value

    def fetch(self):
        out = np.empty((self.n_own, self.C), dtype=self.dtype)
        check(load().glx_dist_sweep_fetch(self._h, _ptr(out)), 'glx_dist_sweep_fetch')
        return out

    def stats(self):
        out = (C.c_int64 * 4)()
        check(load().glx_dist_sweep_stats(self._h, out), 'glx_dist_sweep_stats')
        return dict(sweeps=out[0], exchanges=out[1], graphs=out[2], exchanging=bool(out[3]))

    def info(self):
        """What the object decided (glx_dist_sweep_info): captured / eager exchanging sweeps, the self-test's verdict, ..."""
        out = (C.c_int64 * 8)()
        check(load().glx_dist_sweep_info(self._h, out), 'glx_dist_sweep_info')
        cap = {1: 'captured', 0: 'eager', -1: 'undecided'}[int(out[1])]
        return dict(exchanging=bool(out[0]), exchange=cap if out[0] else 'none', selftest={0: 'not run', 1: 'passed', 2: 'failed'}[int(out[2])],
                    overlap=bool(out[3]), fused=bool(out[4]), scatter=bool(out[5]), send_records=int(out[6]), halo_records=int(out[7]))

    def time_parts(self, reps=50):
        """Microseconds of the rank-local pieces of one sweep, each timed alone (glx_dist_sweep_time_parts)."""
        out = (C.c_float * 4)()
        check(load().glx_dist_sweep_time_parts(self._h, int(reps), out), 'glx_dist_sweep_time_parts')
        return dict(boundary_us=float(out[0]), interior_us=float(out[1]), pack_us=float(out[2]), both_us=float(out[3]))

    # stepwise form (transport left to the caller)
    def begin(self):
        check(load().glx_dist_sweep_begin(self._h), 'glx_dist_sweep_begin')

    def boundary(self, want_err):
        check(load().glx_dist_sweep_boundary(self._h, 1 if want_err else 0), 'glx_dist_sweep_boundary')

    def get_send(self):
        out = np.empty((self.n_send, self.lay['ld']), dtype=self.dtype)
        check(load().glx_dist_sweep_get_send(self._h, _ptr(out)), 'glx_dist_sweep_get_send')
        return out

    def put_halo(self, rec, next_iterate):
        rec = _dense(rec, self.dtype, (self.n_halo, self.lay['ld']), 'halo records')
        check(load().glx_dist_sweep_put_halo(self._h, _ptr(rec), 1 if next_iterate else 0), 'glx_dist_sweep_put_halo')

    def interior(self, want_err):
        e = C.c_double(0)
        check(load().glx_dist_sweep_interior(self._h, 1 if want_err else 0, C.byref(e)), 'glx_dist_sweep_interior')
        return e.value if want_err else None

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib = load(required=False)
            if lib is not None:
                lib.glx_dist_sweep_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def nearest_dist(X, idx, device=None):
    """Distance from every row of X to the nearest of its rows `idx` (glx_nearest_dist: cKDTree(X[idx]).query(X)[0], bit for bit)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise GlxError('nearest_dist: X must be (n, d)')
    idx = np.ascontiguousarray(np.asarray(idx).reshape(-1), dtype=np.int64)
    out = np.empty(X.shape[0], dtype=np.float64)
    check(load().glx_nearest_dist(_ptr(X), X.shape[0], X.shape[1], _ptr(idx), len(idx), _ptr(out), _dev(device)), 'glx_nearest_dist')
    return out


def record_layout(Cc, dtype=np.float64, has_w=True):
    """Vertex-record layout of the dense operands (pure host call, no GPU needed)."""
    out = (C.c_int32 * 6)()
    check(load().glx_record_layout(int(Cc), _dt(dtype), 1 if has_w else 0, out), 'glx_record_layout')
    return dict(ld=out[0], woff=out[1], rec_bytes=out[2], G=out[3], nvec=out[4], esize=out[5])


def argmax_project(prob, priors=None, weights=None, max_steps=0, similarity=True, device=None):
    """ssl.predict / ssl.volume_label_projection on device.
    Returns (labels int64, weights, err, steps)."""
    prob = np.asarray(prob)
    # a float32 prob (use_cuda=True) is normalised in float32 like numpy would (ssl.py:256-257)
    prob = np.ascontiguousarray(prob, dtype=np.float32 if prob.dtype == np.float32 else np.float64)
    n, Cc = prob.shape
    w = np.ones(Cc) if weights is None else np.array(weights, dtype=np.float64).reshape(Cc).copy()
    pri = np.zeros(Cc) if priors is None else _dense(priors, np.float64, (Cc,), 'priors')
    labels = np.empty(n, dtype=np.int64)
    err = C.c_double(0)
    steps = C.c_int(0)
    check(load().glx_argmax_project_t(_ptr(prob), _dt(prob.dtype), n, Cc, _ptr(pri), _ptr(w), _ptr(labels), C.byref(err),
                                      C.byref(steps), int(max_steps), 1 if similarity else 0, _dev(device)), 'glx_argmax_project')
    return labels, w, err.value, steps.value


def lp_iterate(uu, ul, nbr, row, W, ind, val, p, T, tol, device=None):
    """lp_iterate of the reference's C extension on the GPU (in place on uu, ul); returns the stopping iteration."""
    for a, dt in ((uu, np.float64), (ul, np.float64), (nbr, np.int32), (row, np.int32), (W, np.float64), (ind, np.int32), (val, np.float64)):
        if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags['C_CONTIGUOUS']):
            raise GlxError('lp_iterate: arrays must be C-contiguous with the dtypes of the reference binding')
    it = C.c_int64(0)
    check(load().glx_lp_iterate(_ptr(uu), _ptr(ul), _ptr(nbr), _ptr(row), _ptr(W), _ptr(ind), _ptr(val), float(p), int(T), float(tol),
                                len(uu), len(W), len(ind), C.byref(it), _dev(device)), 'glx_lp_iterate')
    return it.value


def host_row_sums(W):
    """W * ones for a CSR matrix with int32 indices (scipy's csr_matvec order), by the library's host loop."""
    out = np.empty(W.shape[0], dtype=np.float64)
    check(load().glx_host_row_sums(W.shape[0], _ptr(W.indptr), _ptr(W.data), _ptr(out)), 'glx_host_row_sums')
    return out


def host_neg_columns_rows(Lcsc, cols, F, row_scale=None):
    """(rows ascending int32, values (len(rows), k)) of -L[:, cols] * F without the rows in `cols`, each row times row_scale[row]
    (glx_host_neg_columns_rows: csr_matvecs' order of a row's terms); None when the preconditions do not hold (the caller then
    takes the literal expression)."""
    n = Lcsc.shape[0]
    cols = np.ascontiguousarray(cols, dtype=np.int64).reshape(-1)
    F = np.ascontiguousarray(F, dtype=np.float64)
    if Lcsc.indptr.dtype != np.int32 or Lcsc.indices.dtype != np.int32 or Lcsc.data.dtype != np.float64 or F.ndim != 2 or len(F) != len(cols):
        return None
    if len(cols) and (cols.min() < 0 or cols.max() >= n):
        return None
    k = F.shape[1]
    cap = int(Lcsc.indptr[cols + 1].astype(np.int64).sum() - Lcsc.indptr[cols].astype(np.int64).sum()) if len(cols) else 0
    rows = np.empty(max(cap, 1), dtype=np.int32)
    vals = np.empty((max(cap, 1), k), dtype=np.float64)
    cnt = C.c_int64(0)
    scale = None if row_scale is None else np.ascontiguousarray(row_scale, dtype=np.float64)
    check(load().glx_host_neg_columns_rows(n, _ptr(Lcsc.indptr), _ptr(Lcsc.indices), _ptr(Lcsc.data), len(cols), _ptr(cols), _ptr(F), k,
                                           _ptr(scale), cap, _ptr(rows), _ptr(vals), C.byref(cnt)), 'glx_host_neg_columns_rows')
    if cnt.value < 0:
        return None
    return rows[:cnt.value], vals[:cnt.value]


def host_reverse_scale_rows(W, scale):
    """(indices, data) of the matrix whose row i is row i of W times scale[i] with the entries in reverse order."""
    col = np.empty(W.nnz, dtype=np.int32)
    val = np.empty(W.nnz, dtype=np.float64)
    check(load().glx_host_reverse_scale_rows(W.shape[0], _ptr(W.indptr), _ptr(W.indices), _ptr(W.data), _ptr(scale), _ptr(col), _ptr(val)),
          'glx_host_reverse_scale_rows')
    return col, val


def auto_cells(n, d):
    """How many cells glx_knn_clustered forms when the caller leaves it to the library: none below 2^17 rows (the all-pairs
    search takes a few ms there) or above 128 features (the cell pruning rides on the split-bf16 filter), else one per 8192
    rows within [16, 256].  GLX_KNN_CLUSTERED=0 turns it off, =<m> forces m cells."""
    e = os.environ.get('GLX_KNN_CLUSTERED')
    if e is not None:
        return max(0, int(e))
    if n < (1 << 17) or d > 128:
        return 0
    return int(min(256, max(64, n // 8192)))      # (>= 64: the cells' order also serves as the vertex order of the graph's operators)


def auto_order_cells(n, d):
    """Below the size of the cell-PRUNED search: how many chained cells the rows are reordered by before the all-pairs search
    (coherent wavefronts: 10-14 % of the search on clustered data, nothing lost elsewhere), whose order the operators on the
    graph then take instead of their own pass over the graph (3.7 ms at 70 000 vertices).  GLX_KNN_ORDER=0 turns it off."""
    if os.environ.get('GLX_KNN_ORDER', '1') == '0' or n < 4096 or n >= (1 << 17) or d > 128:
        return 0
    return int(min(128, n // 64))     # (measured at 70 000 x 20: 128 cells = the library's order to 0.5 %, 64 and 32 cells 0.5-1 % behind)


class _KnnOptions(C.Structure):
    _fields_ = [('filter', C.c_int), ('lists', C.c_int), ('nsplit', C.c_int), ('concat', C.c_int)]


class knn_options:
    """Context manager: plan overrides for the calling thread's kNN searches (glx_knn_set_options) -- which candidate filter
    ('bf16' | 'f32'), list length ('short' | 'long'), ref ranges per query block (1..8), operand form of the bf16 filter at
    d <= 21 (concat 0 | 1 | 2).  Every plan returns the same exact lists; tests and A/B measurements use this."""

    def __init__(self, filter=None, lists=None, nsplit=None, concat=None):
        self.opt = _KnnOptions({None: 0, 'bf16': 1, 'f32': 2}[filter], {None: 0, 'short': 1, 'long': 2}[lists],
                                int(nsplit or 0), -1 if concat is None else int(concat))

    def __enter__(self):
        check(load().glx_knn_set_options(C.byref(self.opt)), 'glx_knn_set_options')
        return self

    def __exit__(self, *exc):
        load().glx_knn_set_options(None)
        return False


def _knn_input(X, similarity):
    X = np.asarray(X, dtype=np.float64)
    if similarity == 'angular':
        X = X / np.linalg.norm(X, axis=1)[:, None]
    elif similarity != 'euclidean':
        raise GlxError('similarity %r not supported (euclidean, angular)' % (similarity,))
    return np.ascontiguousarray(X)


def _knn_cells(n, d, clustered, want_order):
    """The `ncells` argument of glx_knn_clustered / glx_knn_search for a full search."""
    m = auto_cells(n, d) if clustered is None else int(clustered)
    if m <= 1 and clustered is None and want_order and auto_order_cells(n, d) > 1:
        m = -auto_order_cells(n, d)     # all pairs on rows reordered by that many chained cells
    return m if (m > 1 or m < -1) else 0


def knn_bruteforce(X, k, similarity='euclidean', device=None, query_range=None, cell_starts=None, clustered=None, want_order=False):
    """Exact kNN (incl. self) on the GPU.  'angular' = euclidean on row-normalised data, formed
    with the reference's own expression (weightmatrix.py:344-345).  cell_starts: the rows come in a coarse geometric
    order with cell c = rows [cell_starts[c], cell_starts[c+1]); the search skips the cells that cannot hold a
    neighbour (glx_knn_cells_range: the same lists, a fraction of the tiles on clustered data).  clustered: number of cells
    the library forms itself (glx_knn_clustered; None = auto_cells(n, d), 0 = all pairs).
    want_order (below the size of the pruned search): the rows are put into the order of chained cells before the search
    (coherent wavefronts; a plain knnsearch does not ask: 0.1 ms at d = 20, 0.7 ms at d = 128 for 50 000 rows) -- KnnResult
    hands that order out as well."""
    dev_ptr = None
    if hasattr(X, 'data_ptr') and getattr(X, 'is_cuda', False):
        # a torch CUDA tensor (float64, contiguous, euclidean): the search reads it where it is (range / cells forms only)
        if similarity != 'euclidean' or str(X.dtype) != 'torch.float64' or not X.is_contiguous() or (query_range is None and cell_starts is None):
            raise GlxError('a device-resident X must be a contiguous float64 tensor, euclidean, searched by range or by cells')
        n, d = int(X.shape[0]), int(X.shape[1])
        dev_ptr = _vp(X.data_ptr())
    else:
        X = _knn_input(X, similarity)
        n, d = X.shape
    q0, q1 = (0, n) if query_range is None else query_range
    ind = pinned_empty((q1 - q0, k), np.int64)       # page-locked result arrays: the copy back runs at PCIe speed
    dist = pinned_empty((q1 - q0, k), np.float64)
    if cell_starts is not None:
        cs = np.ascontiguousarray(cell_starts, dtype=np.int64)
        check(load().glx_knn_cells_range(dev_ptr or _ptr(X), n, d, k, _ptr(cs), len(cs), q0, q1, _ptr(ind), _ptr(dist), _dev(device)),
              'glx_knn_cells_range')
        return ind, dist
    if dev_ptr is not None:
        check(load().glx_knn_bruteforce_range(dev_ptr, n, d, k, q0, q1, _ptr(ind), _ptr(dist), _dev(device)), 'glx_knn_bruteforce')
        return ind, dist
    m = _knn_cells(n, d, clustered, want_order) if query_range is None else 0
    if m:       # cells formed by the library (same lists; a fraction of the tiles when the data has clusters)
        check(load().glx_knn_clustered(_ptr(X), n, d, k, m, _ptr(ind), _ptr(dist), _dev(device)), 'glx_knn_clustered')
        return ind, dist
    check(load().glx_knn_bruteforce_range(_ptr(X), n, d, k, q0, q1, _ptr(ind), _ptr(dist), _dev(device)),
          'glx_knn_bruteforce')
    return ind, dist


_PINNED_CSR_MAX = 512 << 20      # above this the weight matrix comes back through ordinary memory
_KERNEL_ID = {'given': 0, 'uniform': 1, 'gaussian': 2, 'symgaussian': 3, 'distance': 4, 'singular': 5}


class KnnResult:
    """A finished full search whose lists live on the device (glx_knn_search): `to_csr` builds the weight matrix from them
    without a host round trip, `lists()` copies them out, `order()` is the cell order the search worked out (or None)."""

    def __init__(self, X, k, similarity='euclidean', device=None, clustered=None, want_order=False):
        X = _knn_input(X, similarity)
        n, d = X.shape
        self.n, self.k, self.device = n, int(k), _dev(device)
        self._h = _vp()
        check(load().glx_knn_search(_ptr(X), n, d, int(k), _knn_cells(n, d, clustered, want_order), self.device, C.byref(self._h)),
              'glx_knn_search')

    def lists(self):
        ind = pinned_empty((self.n, self.k), np.int64)
        dist = pinned_empty((self.n, self.k), np.float64)
        check(load().glx_knn_result_lists(self._h, _ptr(ind), _ptr(dist)), 'glx_knn_result_lists')
        return ind, dist

    def order(self):
        perm = pinned_empty((self.n,), np.int32)         # (a device-to-host copy into fresh pageable memory takes milliseconds the first time)
        if load().glx_knn_result_order(self._h, _ptr(perm)) != 0:
            return None
        return perm

    def to_csr(self, k, kernel='gaussian', sym=1, weights=None):
        from scipy import sparse
        n = self.n
        w = None if weights is None else _dense(weights, np.float64, (n, k), 'weights')
        cap = n * int(k) * (2 if sym else 1)
        pinned = cap * 12 <= _PINNED_CSR_MAX
        indptr = pinned_empty((n + 1,), np.int32) if pinned else np.empty(n + 1, np.int32)
        col_buf = pinned_empty((cap,), np.int32) if pinned else np.empty(cap, np.int32)
        val_buf = pinned_empty((cap,), np.float64) if pinned else np.empty(cap, np.float64)
        nnz = C.c_int64(0)
        check(load().glx_knn_result_to_csr(self._h, int(k), _KERNEL_ID[kernel], int(sym), _ptr(w), cap, _ptr(indptr), _ptr(col_buf),
                                           _ptr(val_buf), C.byref(nnz)), 'glx_knn_result_to_csr')
        W = sparse.csr_matrix((val_buf[:nnz.value], col_buf[:nnz.value], indptr), shape=(n, n))
        W.has_sorted_indices = True
        W.has_canonical_format = True
        return W

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            lib = load(required=False)
            if lib is not None:
                lib.glx_knn_result_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exp_cr(x, device=None):
    """exp(x), correctly rounded, evaluated on the device (glx_exp_cr)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    check(load().glx_exp_cr(_ptr(x), _ptr(out), x.size, _dev(device)), 'glx_exp_cr')
    return out


def knn_to_csr(knn_ind, knn_dist, k, kernel='gaussian', sym=1, weights=None, device=None):
    """kNN data -> scipy CSR weight matrix, assembled on the GPU (glx_knn_to_csr)."""
    from scipy import sparse
    ind = np.ascontiguousarray(knn_ind, dtype=np.int64)
    n, kk = ind.shape
    # distances are only read by the kernels that compute weights from them: with given weights they need not travel
    dist = None if (knn_dist is None or kernel == 'given') else _dense(knn_dist, np.float64, (n, kk), 'knn_dist')
    w = None if weights is None else _dense(weights, np.float64, (n, k), 'weights')
    nnz = C.c_int64(0)
    lib = load()
    cap = n * int(k) * (2 if sym else 1)                   # every list entry appears at most twice (as i->j and as j->i)
    if cap * 12 <= _PINNED_CSR_MAX:
        # the CSR lands in page-locked arrays of the pool (recycled by size: the next graph of the same (n, k) reuses them);
        # indices / data are views of the first nnz entries
        indptr = pinned_empty((n + 1,), np.int32)
        col_buf = pinned_empty((cap,), np.int32)
        val_buf = pinned_empty((cap,), np.float64)
        check(lib.glx_knn_to_csr_into(_ptr(ind), _ptr(dist), _ptr(w), n, kk, int(k), _KERNEL_ID[kernel], int(sym), cap, _ptr(indptr),
                                      _ptr(col_buf), _ptr(val_buf), C.byref(nnz), _dev(device)), 'glx_knn_to_csr_into')
        indices, data = col_buf[:nnz.value], val_buf[:nnz.value]
    else:
        rp, ci, va = _vp(), _vp(), _vp()
        check(lib.glx_knn_to_csr(_ptr(ind), _ptr(dist), _ptr(w), n, kk, int(k), _KERNEL_ID[kernel], int(sym), C.byref(rp),
                                 C.byref(ci), C.byref(va), C.byref(nnz), _dev(device)), 'glx_knn_to_csr')
        try:
            m = nnz.value
            indptr = np.ctypeslib.as_array(C.cast(rp, _i32p), shape=(n + 1,)).copy()
            indices = np.ctypeslib.as_array(C.cast(ci, _i32p), shape=(max(m, 1),))[:m].copy()
            data = np.ctypeslib.as_array(C.cast(va, _f64p), shape=(max(m, 1),))[:m].copy()
        finally:
            lib.glx_free(rp)
            lib.glx_free(ci)
            lib.glx_free(va)
    W = sparse.csr_matrix((data, indices, indptr), shape=(n, n))
    W.has_sorted_indices = True
    W.has_canonical_format = True
    return W


def knn_rows_to_csr(J_own, w_own, n_cols, row_base, rev_row, rev_src, rev_pos, rev_w, sym=1, device=None):
    """Rows [row_base, row_base + m) of weightmatrix.knn's matrix from the block's own lists and the reverse entries sent by the
    owners of the other rows, merged on the GPU (glx_knn_rows_to_csr).  Returns a scipy CSR of shape (m, n_cols)."""
    from scipy import sparse
    J = np.ascontiguousarray(J_own, dtype=np.int64)
    m, k = J.shape
    w = _dense(w_own, np.float64, (m, k), 'w_own')
    rr = np.ascontiguousarray(rev_row, dtype=np.int64)
    rs = np.ascontiguousarray(rev_src, dtype=np.int64)
    rp = np.ascontiguousarray(rev_pos, dtype=np.int64)
    rw = np.ascontiguousarray(rev_w, dtype=np.float64)
    cap = m * k + len(rr)
    indptr = np.empty(m + 1, dtype=np.int32)
    col = np.empty(max(cap, 1), dtype=np.int32)
    val = np.empty(max(cap, 1), dtype=np.float64)
    nnz = C.c_int64(0)
    check(load().glx_knn_rows_to_csr(_ptr(J), _ptr(w), m, k, int(n_cols), int(row_base), _ptr(rr), _ptr(rs), _ptr(rp), _ptr(rw), len(rr), int(sym),
                                     cap, _ptr(indptr), _ptr(col), _ptr(val), C.byref(nnz), _dev(device)), 'glx_knn_rows_to_csr')
    W = sparse.csr_matrix((val[:nnz.value].copy(), col[:nnz.value].copy(), indptr), shape=(m, int(n_cols)))
    W.has_sorted_indices = True
    W.has_canonical_format = True
    return W


def host_locality_order(indptr, indices, col_lo=0):
    """perm[new] = old: the library's breadth-first locality order of an n-row pattern restricted to columns [col_lo, col_lo + n)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n = len(indptr) - 1
    perm = np.empty(n, dtype=np.int32)
    check(load().glx_host_locality_order(n, _ptr(indptr), _ptr(indices), int(col_lo), _ptr(perm)), 'glx_host_locality_order')
    return perm


def host_permute_rows(A, perm):
    """csr_matrix with row i = row perm[i] of A (entry order inside the rows kept), on host threads."""
    from scipy import sparse
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    data = np.ascontiguousarray(A.data, dtype=np.float64)
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    n = len(indptr) - 1
    ip, ix, dv = np.empty(n + 1, dtype=np.int32), np.empty(len(indices), dtype=np.int32), np.empty(len(data), dtype=np.float64)
    check(load().glx_host_permute_rows(n, _ptr(indptr), _ptr(indices), _ptr(data), _ptr(perm), _ptr(ip), _ptr(ix), _ptr(dv)), 'glx_host_permute_rows')
    out = sparse.csr_matrix((dv, ix, ip), shape=A.shape)
    out.has_sorted_indices = A.has_sorted_indices
    return out


def knn_stats():
    out = (C.c_double * 16)()
    check(load().glx_knn_stats(out), 'glx_knn_stats')
    return dict(tile_ms=out[0], rerank_ms=out[1], fallback_rows=out[2], total_ms=out[3], fallback_ms=out[4],
                dpa=out[5], nsplit=out[6], KP=abs(out[7]), filter='bf16x3' if out[7] < 0 else 'f32', escalated_rows=out[8],
                concatenated=int(out[9]), seed_sample=int(out[10]), visited_share=out[11], cells=int(out[12]))
