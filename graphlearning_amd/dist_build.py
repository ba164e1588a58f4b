"""Sharded graph build and planning for the vertex-partitioned sweep: O(n/N) graph data per rank.

`dist.py`'s first planner has every rank hold the whole weight matrix and derive every rank's halo.  That is fine at
N x 70 000 vertices and hopeless at config 4 (n = 10^7: 183 M stored entries per rank on the host).  Here every rank
only ever touches its own block of rows [lo, hi) of the ORIGINAL vertex numbering:

  1. kNN lists of its own query rows (glx_knn_bruteforce_range; every rank holds the features -- the one O(n d)
     array the brute-force search needs everywhere);
  2. symmetrisation by owner rank: each list entry (i -> j, w) is also needed by the owner of row j, so the
     entries are routed to their owners with one all-to-all-v of (j, i, w) triples, and every rank assembles ITS
     rows of W = (A + A^T)/2 from its own lists and the triples it received -- with the reference's own scipy
     expressions (weightmatrix.py:170-186) applied to the row block, so the rows are bit-identical to the rows of
     the reference's W;
  3. degrees, D^-1 and its rows of P = D^-1 W^T locally (W is symmetric bit for bit, so row i of W^T is row i of
     W; the product is formed by scipy on the block, which yields the same entry order as the reference's global
     product: the accumulation order of the sweep);
  4. halo planning with a request exchange: the rank lists the remote columns its rows reference (grouped by
     owner = ascending id), sends each owner its request list, and builds its send lists from the requests it
     receives.  No rank computes another rank's halo.

The only O(n) arrays a rank holds are 8-byte-per-vertex vectors (labels, degrees -- `vinf = deg / np.sum(deg)`
needs the global sum in numpy's own summation order to stay bit-identical) and the features.
Functions taking `msgs` work on plain numpy data (unit-testable without processes); `ShardedGraph.build` drives them
through torch.distributed.
"""
import numpy as np
from scipy import sparse

from .dist import block_bounds


# ---- step 0: a cheap locality order before sharding ------------------------------------------------------------------
def coarse_locality_order(X, ncells=64, seed=0, chunk=262144, return_cells=False):
    """Permutation (new -> old) that makes contiguous blocks of vertices geometrically compact BEFORE the graph exists
    ("blobs/clusters first"): every point goes to the nearest of `ncells` sample points, the cells are
    chained greedily by nearest unvisited cell (cells of one cluster end up next to each other), and points are
    sorted by their cell's place in the chain (stable: original order inside a cell).  O(n * ncells * d) on the host,
    the same on every rank.  Ranks that own contiguous blocks of this order import only the neighbours across their
    block boundaries instead of (for data in arbitrary order) nearly every vertex of the graph.
    return_cells: also the positions (in the new order) at which a new cell starts -- the natural candidates for block
    boundaries (graph_cut_bounds): a cut between two cells of different clusters crosses no edge at all."""
    X = np.asarray(X)
    n = X.shape[0]
    ncells = int(min(ncells, n))
    rng = np.random.default_rng(seed)
    cent = X[np.sort(rng.choice(n, size=ncells, replace=False))].astype(np.float64)
    cn = np.einsum('ij,ij->i', cent, cent)
    cell = np.empty(n, dtype=np.int32)
    for lo in range(0, n, chunk):
        blk = X[lo:lo + chunk]
        d2 = cn[None, :] - 2.0 * (blk @ cent.T)            # + |x|^2, constant per row
        cell[lo:lo + chunk] = np.argmin(d2, axis=1)
    # chain of cells: start at the cell farthest from the centroid mean, always go to the nearest unvisited one
    dc = cn[:, None] + cn[None, :] - 2.0 * (cent @ cent.T)
    start = int(np.argmax(np.sum((cent - cent.mean(axis=0)) ** 2, axis=1)))
    place = np.full(ncells, -1, dtype=np.int64)
    cur, seen = start, np.zeros(ncells, dtype=bool)
    for pos in range(ncells):
        place[cur] = pos
        seen[cur] = True
        if pos + 1 < ncells:
            cand = np.where(seen, np.inf, dc[cur])
            cur = int(np.argmin(cand))
    key = place[cell]
    perm = np.argsort(key, kind='stable').astype(np.int64)
    if not return_cells:
        return perm
    starts = np.concatenate([[0], np.cumsum(np.bincount(key, minlength=ncells))[:-1]]).astype(np.int64)
    return perm, starts


def coarse_locality_order_torch(Xt, ncells=64, seed=0, chunk=1 << 20):
    """coarse_locality_order with the O(n * ncells * d) part on the device: Xt is an (n, d) float64 tensor on the GPU.  The same
    sample points, the same chain, the same stable sort; returns (perm tensor new -> old on Xt's device, cell starts as a host
    array).  (A point whose two nearest sample points are equidistant to the last bit may land in the other cell than on the
    host: any assignment is a valid locality order as long as every rank computes the same one, which identical GPUs do.)"""
    import torch
    n = Xt.shape[0]
    ncells = int(min(ncells, n))
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(n, size=ncells, replace=False))
    cent_t = Xt[torch.from_numpy(pick).to(Xt.device)]
    cent = cent_t.cpu().numpy().astype(np.float64)
    cn = np.einsum('ij,ij->i', cent, cent)
    cn_t = torch.from_numpy(cn).to(Xt.device)
    cell = torch.empty(n, dtype=torch.int64, device=Xt.device)
    for lo in range(0, n, chunk):
        blk = Xt[lo:lo + chunk]
        cell[lo:lo + chunk] = torch.argmin(cn_t[None, :] - 2.0 * (blk @ cent_t.T), dim=1)
    dc = cn[:, None] + cn[None, :] - 2.0 * (cent @ cent.T)
    start = int(np.argmax(np.sum((cent - cent.mean(axis=0)) ** 2, axis=1)))
    place = np.full(ncells, -1, dtype=np.int64)
    cur, seen = start, np.zeros(ncells, dtype=bool)
    for pos in range(ncells):
        place[cur] = pos
        seen[cur] = True
        if pos + 1 < ncells:
            cur = int(np.argmin(np.where(seen, np.inf, dc[cur])))
    key = torch.from_numpy(place).to(Xt.device)[cell]
    perm = torch.sort(key, stable=True).indices
    counts = torch.bincount(key, minlength=ncells).cpu().numpy()
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    return perm, starts


# ---- step 1b: block boundaries that follow the graph ---------------------------------------------------------------
def crossing_counts_at(J_own, lo, cand):
    """How many kNN list entries (i -> j) of the rows [lo, lo + len(J_own)) cross a cut placed at each candidate position
    (a cut at p separates vertices < p from vertices >= p): the local part of the count, to be summed over the ranks."""
    cand = np.asarray(cand, dtype=np.int64)
    J = np.asarray(J_own, dtype=np.int64)
    i = np.repeat(np.arange(lo, lo + J.shape[0], dtype=np.int64), J.shape[1])
    j = J.reshape(-1)
    a, b = np.minimum(i, j), np.maximum(i, j)
    first = np.searchsorted(cand, a, side='right')          # candidates p with a < p ...
    last = np.searchsorted(cand, b, side='right')           # ... and p <= b
    diff = np.bincount(first, minlength=len(cand) + 1) - np.bincount(last, minlength=len(cand) + 1)
    return np.cumsum(diff)[:len(cand)].astype(np.int64)


def graph_cut_bounds(dist, n, J_own, lo, cell_starts, group=None, device=None, slack_levels=(0.0, 0.1, 0.25, 0.45, 0.65)):
    """Block boundaries for the sharded pipeline that follow the graph (what dist.cut_bounds does with the global matrix,
    here from the ranks' own kNN lists): candidates are the equal-split points and the cell starts of the coarse geometric
    order; the entries crossing each are counted locally and summed over the ranks (one small all-reduce); the cuts of least
    crossing within a growing imbalance allowance are chosen by the same dynamic programme.  For clustered data the cuts land
    between clusters -- a rank that owns whole clusters has NO halo -- where equal blocks cut through a cluster and, the kNN
    graph of an isotropic cluster being expander-like, import about as many rows as they own (profiles/r03_scale_model.json:
    n = 2e6, 8 equal blocks: 140 000 halo rows per 250 000 owned, one xGMI link carrying 18 MB per sweep).
    Every rank returns the same bounds."""
    import torch
    from . import dist as gdist
    world = dist.get_world_size(group)
    if world <= 1:
        return np.array([0, n], dtype=np.int64)
    cand = sorted({int(c) for c in list(cell_starts) + [(n * r) // world for r in range(1, world)] if 0 < int(c) < n})
    xc = torch.from_numpy(crossing_counts_at(J_own, lo, cand))
    if dist.get_backend(group) == 'nccl':
        dev = torch.device('cuda', device) if isinstance(device, int) else (device if device is not None else torch.device('cuda', torch.cuda.current_device()))
        xc = xc.to(dev)
    dist.all_reduce(xc, group=group)
    return gdist.choose_cuts(xc.cpu().numpy(), cand, n, world, slack_levels)


def redistribute_rows(dist, arrays, old_bounds, new_bounds, group=None, device=None):
    """Rows of 2-D arrays (kNN lists: J int64, D float64) that the ranks hold for the contiguous blocks `old_bounds` are moved
    to the owners under `new_bounds` (one all-to-all-v per array).  Returns the arrays of this rank's new block."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo = int(old_bounds[rank])
    out = []
    for A in arrays:
        A = np.ascontiguousarray(A)
        ncol = A.shape[1]
        parts = []
        for r in range(world):
            a = min(max(int(new_bounds[r]) - lo, 0), A.shape[0])
            b = min(max(int(new_bounds[r + 1]) - lo, 0), A.shape[0])
            parts.append(A[a:b].reshape(-1))
        dtype = np.int64 if A.dtype == np.int64 else np.float64
        got = _alltoallv(dist, parts, dtype, group, device)
        out.append(np.concatenate(got).reshape(-1, ncol))        # sources arrive in rank order = ascending row
    return out


# ---- step 2: symmetrisation by owner -------------------------------------------------------------------------------
def knn_weights_rows(J, D, k, kernel='gaussian'):
    """Kernel weights of a block of kNN lists (reference weightmatrix.py:134-156; every kernel here needs only the
    row's own distances).  k counts the self point."""
    k = int(min(J.shape[1], k))
    J, D = J[:, :k], D[:, :k]
    if kernel == 'uniform':
        w = np.ones_like(D)
    elif kernel == 'gaussian':
        from .weightmatrix import _row_blocks
        w = np.empty(D.shape, dtype=np.float64)

        def rows(lo, hi):           # elementwise, so any split into row blocks gives the same bits (host threads: numpy frees the GIL)
            sq = D[lo:hi] * D[lo:hi]
            np.exp(-4 * sq / sq[:, k - 1][:, None], out=w[lo:hi])
        _row_blocks(rows, D.shape[0])
    elif kernel == 'distance':
        w = D
    elif kernel == 'singular':
        w = D.copy()
        w[D == 0] = 1
        w = 1 / w
    else:
        raise ValueError("kernel %r needs remote rows' bandwidths; the sharded build supports uniform, gaussian, distance, singular" % (kernel,))
    return J, w


def reverse_messages(J, w, lo, bounds):
    """For the list entries (i -> j, w_ij) of the rows [lo, lo+len(J)): the tuples (j, i, w_ij, position of j in row i's list)
    the owner of row j needs, per destination rank, in list order (row after row, neighbour after neighbour)."""
    n_rows, k = J.shape
    rows_i = np.repeat(np.arange(lo, lo + n_rows, dtype=np.int64), k)
    cols_j = J.reshape(-1).astype(np.int64)
    vals = np.ascontiguousarray(w, dtype=np.float64).reshape(-1)
    pos = np.tile(np.arange(k, dtype=np.int64), n_rows)
    world = len(bounds) - 1
    if world == 1:
        return [(cols_j, rows_i, vals, pos)]
    dest = (np.searchsorted(bounds, cols_j, side='right') - 1).astype(np.int8 if world < 128 else np.int32)
    out = []
    for r in range(world):                                       # one pass per destination (few): list order survives, no sort of n k keys
        sel = np.flatnonzero(dest == r)
        out.append((cols_j[sel], rows_i[sel], vals[sel], pos[sel]))
    return out


def assemble_rows_device(lo, hi, n, J, w, received, symmetrize=True, sym_rule='mean', device=None):
    """assemble_rows on the GPU (glx_knn_rows_to_csr: the merge kernels of the single-GPU assembly applied to the block):
    bit-identical rows, without the scipy COO / CSR passes over the block (14.7 s at one rank of n = 1e7).  `received`: tuples
    (j, i, w_ij, pos) per source rank."""
    from . import _hip
    if symmetrize and received:
        cat = (lambda q: received[0][q]) if len(received) == 1 else (lambda q: np.concatenate([t[q] for t in received]))   # (one source: no 0.9 GB copies)
        rj, ri, rv, rp = cat(0), cat(1), cat(2), cat(3)
    else:
        rj = ri = rp = np.zeros(0, np.int64)
        rv = np.zeros(0)
    sym = 0 if not symmetrize else (1 if sym_rule == 'mean' else 2)
    return _hip.knn_rows_to_csr(J, w, n, lo, rj, ri, rp, rv, sym=sym, device=device)


def device_assembly_available():
    """Is there a GPU (and the library) to assemble on?  The CPU tests run the scipy expressions."""
    from . import _hip
    try:
        return _hip.load(required=False) is not None and _hip.device_count() > 0
    except Exception:
        return False


def assemble_rows(lo, hi, n, J, w, received, symmetrize=True, sym_rule='mean'):
    """Rows [lo, hi) of the weight matrix of weightmatrix.knn from the block's own lists (J, w) and the reverse tuples
    `received` (list over source ranks, in rank order, of (j, i, w_ij[, pos]) with j in [lo, hi)).  The reference's expressions
    (weightmatrix.py:170-186) on the row block: COO -> CSR sums duplicates, (W + W^T)/2 or the element-wise max,
    zero diagonal, explicit zeros dropped."""
    m = hi - lo
    k = J.shape[1]
    rows = (np.ones((m, k)) * np.arange(m)[:, None]).flatten()           # float row ids like weightmatrix.py:171
    A = sparse.coo_matrix((np.asarray(w).flatten(), (rows, J.flatten())), shape=(m, n)).tocsr()
    if symmetrize:
        if received:
            rj = np.concatenate([t[0] for t in received]) - lo
            ri = np.concatenate([t[1] for t in received])
            rv = np.concatenate([t[2] for t in received])
        else:
            rj, ri, rv = np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
        # row j of A^T = the entries (i -> j): ordered by source row i like the transpose of a canonical CSR
        AT = sparse.coo_matrix((rv, (rj, ri)), shape=(m, n)).tocsr()
        if sym_rule == 'mean':
            W = (A + AT) / 2
        elif sym_rule == 'max':                                           # utils.sparse_max (utils.py:263-286)
            nz = (A + AT) > 0
            b_wins = AT > A
            a_wins = nz - b_wins
            W = A.multiply(a_wins) + AT.multiply(b_wins)
        else:
            raise ValueError(sym_rule)
    else:
        W = A
    W = sparse.csr_matrix(W)
    # setdiag(0) + eliminate_zeros on the block: the diagonal of row r is column lo + r
    rr = np.repeat(np.arange(m), np.diff(W.indptr))
    W.data[W.indices == rr + lo] = 0
    W.eliminate_zeros()
    W.sort_indices()
    return W


# ---- step 3: rows of the Poisson operator --------------------------------------------------------------------------
def poisson_rows(W_own):
    """deg, D^-1 and the rows of P = D^-1 W^T for a block of rows of a SYMMETRIC W (ssl.py:615-617, 634-635): row i of
    W^T is row i of W bit for bit, and `D * rows` through scipy's csr product leaves every row's entries in the order
    the reference's global product leaves them (the accumulation order of `P*u`)."""
    m, n = W_own.shape
    from . import _hip
    plain = (W_own.indptr.dtype == np.int32 and W_own.indices.dtype == np.int32 and W_own.data.dtype == np.float64
             and W_own.data.flags.c_contiguous and W_own.indices.flags.c_contiguous and W_own.indptr.flags.c_contiguous)
    if plain and _hip.load(required=False) is not None:
        # the same arrays by the library's host loops (ssl._poisson_operator_symmetric: row sums in stored order, row i of W scaled
        # by 1/deg_i with its entries reversed -- what scipy's csr product emits), without scipy's passes over 10^8 entries
        deg = _hip.host_row_sums(W_own)
        dinv = deg ** (-1)
        indices, data = _hip.host_reverse_scale_rows(W_own, dinv)
        P = sparse.csr_matrix((data, indices, W_own.indptr.copy()), shape=(m, n))
        P.has_sorted_indices = False
        return P, deg, sparse.spdiags(dinv, 0, m, m).tocsr()
    deg = W_own * np.ones(n)                                               # graph.degree_vector: csr row sums in stored order
    D = sparse.spdiags(deg ** (-1), 0, m, m).tocsr()                      # graph.degree_matrix(p=-1): d**p
    P = D * W_own
    return sparse.csr_matrix(P), deg, D


# ---- step 4: halo planning with a request exchange -----------------------------------------------------------------
def halo_requests(P_own, lo, hi, bounds):
    """The remote columns the block's rows reference: (needed ids ascending = grouped by owner, per-owner request lists)."""
    if len(bounds) == 2 and lo == bounds[0] and hi == bounds[1]:      # one rank owns every row: nothing to import
        return np.zeros(0, dtype=np.int64), [np.zeros(0, dtype=np.int64)]
    seen = np.zeros(P_own.shape[1], dtype=bool)                 # distinct columns by a mark pass, not by sorting nnz keys
    seen[P_own.indices] = True
    seen[lo:hi] = False
    needed = np.flatnonzero(seen).astype(np.int64)
    owner = np.searchsorted(bounds, needed, side='right') - 1
    reqs = [needed[owner == r] for r in range(len(bounds) - 1)]
    return needed, reqs


class ShardPlan:
    """The fields dist.RankPlan has, built from the rank's own rows and the requests its peers sent (no global matrix)."""

    def __init__(self, P_own, lo, hi, n, rank, bounds, needed, requests_from_peers, global_halo, local_order='block'):
        """local_order: 'block' keeps the block's own order inside the boundary rows and inside the interior rows; 'rcm' sorts each
        of the two groups by the library's breadth-first locality order of the block's rows (links among the rank's own rows; the
        rectangular rank-local operator is never renumbered by the library itself).  Results do not depend on it -- a row's entries
        keep their order.  (On rows that already come in chained cells of feature space it buys nothing: 253 us per sweep either way.)"""
        world = len(bounds) - 1
        self.rank, self.world, self.n_global = rank, world, n
        m = hi - lo
        self.n_own = m
        self.halo = needed
        self.n_halo = len(needed)
        owner = np.searchsorted(bounds, needed, side='right') - 1
        self.recv_counts = [int(np.sum(owner == r)) for r in range(world)]
        send = [np.asarray(requests_from_peers[r], dtype=np.int64) - lo if r != rank else np.zeros(0, np.int64) for r in range(world)]
        self.send_counts = [len(x) for x in send]
        send_idx = np.concatenate(send) if send else np.zeros(0, np.int64)
        self.global_halo = int(global_halo)
        is_b = np.zeros(m, dtype=bool)
        is_b[send_idx] = True
        bnd, inner = np.flatnonzero(is_b), np.flatnonzero(~is_b)
        if local_order == 'rcm' and m > 0:
            from . import _hip
            rank_of = np.empty(m, dtype=np.int64)
            rank_of[_hip.host_locality_order(P_own.indptr, P_own.indices, col_lo=lo)] = np.arange(m)
            bnd, inner = bnd[np.argsort(rank_of[bnd], kind='stable')], inner[np.argsort(rank_of[inner], kind='stable')]
        elif local_order != 'block':
            raise ValueError('local_order must be block or rcm')
        perm_local = np.concatenate([bnd, inner])                                        # boundary rows first
        new_of_old = np.empty(m, dtype=np.int64)
        new_of_old[perm_local] = np.arange(m)
        self.own = lo + perm_local                                                       # global ids in local order
        self.send_idx = new_of_old[send_idx]
        self.n_boundary = int(is_b.sum())
        if np.array_equal(perm_local, np.arange(m)):                                     # the local order is the block's order: no row shuffle
            sub = P_own
        else:
            from . import _hip
            if _hip.load(required=False) is not None:
                sub = _hip.host_permute_rows(P_own, perm_local)                          # host threads of libglx; each row's entry order kept
            else:
                sub = sparse.csr_matrix(P_own[perm_local, :])                            # (scipy's row slicing does the same, slowly)
        cols = sub.indices
        if sub is P_own and len(needed) == 0 and lo == 0 and cols.dtype == np.int32:
            local = cols                                                # one block in its own order: the columns are local already
        else:
            local_of = np.full(n, -1, dtype=np.int64)                   # global id -> local column: a table, not a binary search per entry
            local_of[lo:hi] = new_of_old
            local_of[needed] = m + np.arange(len(needed), dtype=np.int64)
            local = local_of[cols].astype(np.int32)
        self.P_local = sparse.csr_matrix((sub.data, local, sub.indptr), shape=(m, m + self.n_halo))
        self.P_local.has_sorted_indices = False


# ---- driver over torch.distributed ---------------------------------------------------------------------------------
def _alltoallv(dist, arrays, dtype, group=None, device=None):
    """Variable all-to-all of 1-D numpy arrays (one per destination) -> list per source."""
    import torch
    world = dist.get_world_size(group)
    if world == 1:                                   # nothing to move
        return [np.ascontiguousarray(arrays[0], dtype=dtype)]
    tdt = {np.int64: torch.int64, np.float64: torch.float64}[dtype]
    counts = torch.tensor([len(a) for a in arrays], dtype=torch.int64)
    rcounts = torch.empty(world, dtype=torch.int64)
    dev = None
    if dist.get_backend(group) == 'nccl':          # device collectives: tensors live on this rank's GPU
        dev = torch.device('cuda', device) if isinstance(device, int) else (device if device is not None else torch.device('cuda', torch.cuda.current_device()))
    if dev is not None:
        counts, rcounts = counts.to(dev), rcounts.to(dev)
    dist.all_to_all_single(rcounts, counts, group=group)
    rc = [int(x) for x in rcounts.cpu()]
    send = torch.from_numpy(np.ascontiguousarray(np.concatenate(arrays) if arrays else np.zeros(0), dtype=dtype))
    recv = torch.empty(sum(rc), dtype=tdt)
    if dev is not None:
        send, recv = send.to(dev), recv.to(dev)
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=[len(a) for a in arrays], group=group)
    out = recv.cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(rc)])
    return [out[offs[r]:offs[r + 1]] for r in range(world)]


class ShardedGraph:
    """One rank's rows of the kNN weight matrix and of the Poisson operator plus its exchange plan, built collectively."""

    def __init__(self, dist, n, J_own, D_own, k, kernel='gaussian', symmetrize=True, group=None, device=None, bounds=None, assemble='auto',
                 local_order='block'):
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        self.dist, self.group, self.rank, self.world, self.n = dist, group, rank, world, n
        self.bounds = bounds = block_bounds(n, world) if bounds is None else np.asarray(bounds, dtype=np.int64)   # any contiguous blocks
        self.lo, self.hi = lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        assert J_own.shape[0] == hi - lo
        J, w = knn_weights_rows(np.asarray(J_own), np.asarray(D_own), k + 1, kernel)
        sym_rule = 'max' if kernel in ('distance', 'uniform', 'singular') else 'mean'
        received = None
        if symmetrize:
            msgs = reverse_messages(J, w, lo, bounds)
            rj = _alltoallv(dist, [m[0] for m in msgs], np.int64, group, device)
            ri = _alltoallv(dist, [m[1] for m in msgs], np.int64, group, device)
            rv = _alltoallv(dist, [m[2] for m in msgs], np.float64, group, device)
            rp = _alltoallv(dist, [m[3] for m in msgs], np.int64, group, device)
            received = list(zip(rj, ri, rv, rp))
        # the merge of the block's rows: on the GPU when there is one (assemble='device' / 'host' force a path)
        on_device = assemble == 'device' or (assemble == 'auto' and device_assembly_available())
        self.assembled_on = 'device' if on_device else 'host'
        if on_device:
            dev_idx = device.index if hasattr(device, 'index') and not isinstance(device, int) else device
            self.W_own = assemble_rows_device(lo, hi, n, J, w, received, symmetrize, sym_rule, device=dev_idx)
        else:
            self.W_own = assemble_rows(lo, hi, n, J, w, received, symmetrize, sym_rule)
        self.P_own, self.deg_own, self.D_own = poisson_rows(self.W_own)
        needed, reqs = halo_requests(self.P_own, lo, hi, bounds)
        got = _alltoallv(dist, reqs, np.int64, group, device)
        import torch
        tot = torch.tensor([len(needed)], dtype=torch.int64)
        if dist.get_backend(group) == 'nccl':
            tot = tot.to(torch.device('cuda', device) if isinstance(device, int) else (device if device is not None else torch.device('cuda', torch.cuda.current_device())))
        dist.all_reduce(tot, group=group)
        self.plan = ShardPlan(self.P_own, lo, hi, n, rank, bounds, needed, got, int(tot.item()), local_order=local_order)

    def all_degrees(self):
        """deg of every vertex on every rank (8 bytes per vertex): `vinf = deg / np.sum(deg)` needs numpy's own sum."""
        parts = [None] * self.world
        self.dist.all_gather_object(parts, self.deg_own, group=self.group)
        return np.concatenate(parts)

    def poisson_problem_rows(self, train_ind, train_labels):
        """This rank's rows (local order) of Db, w0, deg, vinf for ssl.poisson(gradient_descent) (ssl.py:619-622, 636-644)."""
        from . import utils
        train_ind = np.asarray(train_ind)
        k = len(np.unique(train_labels))
        onehot = utils.labels_to_onehot(np.asarray(train_labels), k)
        rows_src = onehot - np.mean(onehot, axis=0)
        deg_all = self.all_degrees()
        vinf_all = deg_all / np.sum(deg_all)
        own = self.plan.own
        Db = np.zeros((len(own), k))
        w0 = np.zeros(len(own))
        vval = 1.0 / float(len(train_ind))
        mine = np.flatnonzero((train_ind >= self.lo) & (train_ind < self.hi))
        if len(mine):
            pos = np.empty(self.hi - self.lo, dtype=np.int64)      # local position of every owned vertex (two vector passes, no dict of 10^7 keys)
            pos[own - self.lo] = np.arange(len(own), dtype=np.int64)
            for q in mine:
                t = int(train_ind[q])
                p = int(pos[t - self.lo])
                Db[p] = (deg_all[t] ** (-1)) * rows_src[q]      # row of D*source: one product per entry
                w0[p] = vval / deg_all[t]
        return dict(Db=Db, w0=w0, deg=deg_all[own], vinf=vinf_all[own], k=k, deg_all=deg_all, vinf_all=vinf_all)


def poisson_fit_sharded(dist, n, J_own, D_own, k, train_ind, train_labels, engine='glx', ops_factory=None, comm=None, device=None,
                        min_iter=50, max_iter=1000, kernel='gaussian', group=None, check_every=8, gather=True, dtype=np.float64, bounds=None,
                        local_order='block'):
    """weightmatrix.knn + ssl.poisson(solver='gradient_descent').fit with every rank holding only its block of rows:
    (J_own, D_own) are the kNN lists (self included, k+1 columns) of the rank's rows [lo, hi) of `bounds` (default block_bounds(n, world)).
    engine 'glx': the library-owned sweep (glx_dist_sweep over a libglx RCCL communicator); 'glxstep': the same object with
    `dist` moving the packed records (ranks sharing one GPU); 'ops': dist.DistSweep with the rank-local kernel from
    ops_factory(plan, classes) (CPU tests).  Returns (u, T, sharded graph); u is the full
    (n, C) matrix (gather=True) or this rank's rows in plan.own order."""
    from . import dist as gdist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sg = ShardedGraph(dist, n, J_own, D_own, k, kernel=kernel, group=group, device=device, bounds=bounds, local_order=local_order)
    prob = sg.poisson_problem_rows(train_ind, train_labels)
    plan = sg.plan
    err0 = 0.0
    if min_iter == 0:
        v = np.zeros(n)
        v[np.asarray(train_ind)] = 1.0 / float(len(train_ind))
        err0 = float(np.max(np.absolute(v - prob['vinf_all'])))
    if engine == 'glxstep':          # the C-ABI sweep object with `dist` as the transport (several ranks on ONE GPU, or no RCCL)
        from . import _hip
        comm_s = _hip.Comm(world, rank, None, device)
        ds = gdist.glx_dist_sweep(comm_s, plan, prob['k'], dtype=dtype)
        ds.set_problem(prob['Db'], prob['w0'], prob['deg'], prob['vinf'])
        T = gdist.run_stepwise(ds, plan, dist, min_iter, max_iter, err0, group)
        u_own = ds.fetch()
        ds.close()
        comm_s.close()
    elif engine == 'glx':
        own_comm = comm is None
        if comm is None:
            comm = gdist.init_comm(dist, device, group)
        ds = gdist.glx_dist_sweep(comm, plan, prob['k'], dtype=dtype, force_exchange=gdist._force_collectives())
        ds.set_problem(prob['Db'], prob['w0'], prob['deg'], prob['vinf'])
        T, _ = ds.run(min_iter, max_iter, check_every, err0)
        u_own = ds.fetch()
        ds.close()
        if own_comm:
            comm.close()
    else:
        ops = ops_factory(plan, prob['k'])
        sweep = gdist.DistSweep(plan, ops, dist, group)
        sweep.setup(prob['Db'], prob['w0'], prob['deg'], prob['vinf'])
        T = sweep.run(min_iter, max_iter, err0 if min_iter == 0 else None)
        u_own = sweep.result_own()
        sweep.close()
        if hasattr(ops, 'close'):
            ops.close()
    if not gather:
        return u_own, T, sg
    parts = [None] * world
    dist.all_gather_object(parts, (plan.own, u_own), group=group)
    u = np.zeros((n, prob['k']), dtype=u_own.dtype)
    for ids, block in parts:
        u[ids] = block
    return u, T, sg
