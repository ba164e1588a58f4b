"""Vertex-partitioned Poisson sweep across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  The reference has
no distributed code: this is new functionality around the same sweep
(reference graphlearning/ssl.py:631-670).

  * the vertices are reordered for locality (reverse Cuthill-McKee on the symmetrised pattern)
    and cut into `world` contiguous row blocks; rank r owns block r of P, Db, u;
  * rank-local columns are renumbered [owned | halo], the halo being the distinct remote rows
    its block references, grouped by owner rank;
  * the rank's rows are ordered boundary-first; every sweep: local sliced-ELL SpMM of the
    boundary rows (HIP kernel on the rank's torch stream), then ONE all_to_all_single of the
    boundary vertex records straight into the peers' halo regions (xGMI is point-to-point: a
    direct exchange uses all links at once; no ring) while the SpMM of the interior rows runs;
  * the stop column rides inside the vertex record; the stop test is a scalar all_reduce(MAX),
    evaluated from sweep min_iter on.
The ENTRY order inside every row is untouched by the partitioning, so the distributed iterates
are bit-identical to the single-GPU ones and to the reference's CPU path.

The partition / halo / exchange logic is backend-agnostic (`ops` object): `HipOps` runs the
rank-local sweep through libglx on torch CUDA tensors; the CPU tests inject a scipy-backed
`ops` and run the same driver over the gloo backend with world_size 2.
"""
import numpy as np
from scipy import sparse


def locality_order(P):
    """Permutation (new -> old) that keeps graph neighbours close: reverse Cuthill-McKee on
    the symmetrised sparsity pattern."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    A = sparse.csr_matrix(P)
    pattern = sparse.csr_matrix((np.ones(A.nnz, dtype=np.int8), A.indices, A.indptr), shape=A.shape)
    pattern = (pattern + pattern.T).tocsr()
    return np.asarray(reverse_cuthill_mckee(pattern, symmetric_mode=True), dtype=np.int64)


def block_bounds(n, world):
    """`world` contiguous blocks of (nearly) equal size."""
    return np.array([(n * r) // world for r in range(world + 1)], dtype=np.int64)


def crossing_counts(P, order):
    """cross[p] = number of stored entries (i, j) of the symmetrised pattern whose endpoints lie on
    different sides of a cut between positions p-1 and p of `order` (cross[0] = cross[n] = 0)."""
    A = sparse.csr_matrix(P)
    n = A.shape[0]
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    coo = A.tocoo()
    a = np.minimum(pos[coo.row], pos[coo.col])
    b = np.maximum(pos[coo.row], pos[coo.col])
    diff = np.bincount(a + 1, minlength=n + 2) - np.bincount(b + 1, minlength=n + 2)
    return np.cumsum(diff)[:n + 1]


def _best_cuts(xc, n, world, cap, cand):
    """world-1 increasing cut positions among `cand` (xc[i] = entries crossing a cut at cand[i]) with every block <= cap
    rows, minimising the total crossing count, ties towards balanced blocks (least sum of squared block sizes, then the
    earliest predecessor).  Returns (total crossing, cuts) or None.  Dynamic programme over (number of
    cuts placed, last cut), one numpy pass per cut: O(world * len(cand)^2) array work."""
    c = np.asarray(cand, dtype=np.int64)
    m = len(c)
    if m == 0:
        return None
    BIG = np.iinfo(np.int64).max // 4
    xc = np.asarray(xc).astype(np.int64)
    ok0 = c <= cap
    cost = np.where(ok0, xc, BIG)                         # crossings of the cuts placed so far, last cut at c[j]
    bal = np.where(ok0, c * c, BIG)                       # sum of squared sizes of the blocks closed so far
    gap = c[None, :] - c[:, None]                         # gap[i, j] = c[j] - c[i]
    allowed = (gap > 0) & (gap <= cap)
    back = []
    for _ in range(2, world):
        feas = allowed & (cost < BIG)[:, None]
        t0 = np.where(feas, cost[:, None] + xc[None, :], BIG)
        best0 = t0.min(axis=0)
        t1 = np.where(feas & (t0 == best0[None, :]), bal[:, None] + gap * gap, BIG)
        arg = t1.argmin(axis=0)                           # first minimiser = earliest predecessor, like the scalar loop
        best1 = t1[arg, np.arange(m)]
        none = best0 >= BIG
        cost = np.where(none, BIG, best0)
        bal = np.where(none, BIG, best1)
        back.append(np.where(none, -1, arg))
    last = n - c
    feas = (cost < BIG) & (last <= cap)
    if not np.any(feas):
        return None
    best0 = np.where(feas, cost, BIG).min()
    t1 = np.where(feas & (cost == best0), bal + last * last, BIG)
    bj = int(t1.argmin())
    cuts = [int(c[bj])]
    for arg in reversed(back):
        bj = int(arg[bj])
        cuts.append(int(c[bj]))
    return int(best0), cuts[::-1]


def cut_bounds(P, order, world, slack_levels=(0.0, 0.1, 0.25, 0.45, 0.65, 0.85)):
    """Block boundaries along `order` that follow the graph: a kNN graph of clustered data is a set of
    (nearly) disconnected pieces which the locality order lays out one after another, and a boundary
    placed in the gap between two pieces costs no halo at all -- a rank that owns whole pieces has
    nothing to exchange.  For growing allowed imbalance (block size <= (1 + slack) n/world) the cuts
    of least total crossing are found by dynamic programming over the low-crossing positions; a larger
    slack is accepted only if it removes the crossings entirely or cuts them at least four-fold, so a
    graph without such structure keeps equal blocks (slack 0 = block_bounds)."""
    n = len(order)
    if world <= 1:
        return np.array([0, n], dtype=np.int64)
    cross = crossing_counts(P, order)
    ideal = [(n * r) // world for r in range(1, world)]
    # candidate positions: the equal-split points and the least-crossed position of each of 32*world windows
    cand = set(ideal)
    edges = np.linspace(1, n, 32 * world + 1).astype(np.int64)
    for lo, hi in zip(edges[:-1], edges[1:]):
        if hi > lo:
            cand.add(int(lo + np.argmin(cross[lo:hi])))
    cand = sorted(c for c in cand if 0 < c < n)
    return choose_cuts(cross[np.asarray(cand, dtype=np.int64)] if cand else np.zeros(0, np.int64), cand, n, world, slack_levels)


def choose_cuts(xc, cand, n, world, slack_levels=(0.0, 0.1, 0.25, 0.45, 0.65, 0.85)):
    """Block boundaries among the candidate positions `cand` (xc[i] = entries crossing a cut at cand[i]): for growing allowed
    imbalance the cuts of least total crossing; a larger slack is accepted only if it removes the crossings entirely or
    cuts them at least four-fold (cut_bounds).  Falls back to equal blocks."""
    chosen = None
    for slack in slack_levels:
        cap = int(np.ceil((1.0 + slack) * n / world))
        res = _best_cuts(xc, n, world, cap, cand)
        if res is None:
            continue
        if chosen is None or res[0] == 0 or 4 * res[0] <= chosen[0]:
            if chosen is None or res[0] < chosen[0]:
                chosen = res
        if chosen is not None and chosen[0] == 0:
            break
    if chosen is None:
        return block_bounds(n, world)
    return np.array([0] + list(chosen[1]) + [n], dtype=np.int64)


def partition_cost(P, order, bounds):
    """What a partition costs, rank by rank, without building the plans: stored entries and rows owned, distinct halo rows
    imported, the largest message from one peer.  Plus a coarse estimate of the time of one fused sweep (the constants of
    glx_dist_sweep_create's own choice between its forms: 2.5 us + 7.7 ps per stored entry for the rows, 6.6 us + 128-byte records
    at 50 GB/s for the largest per-peer message as soon as ANY rank imports a halo) -- enough to rank partitions against each
    other; scripts/scale_model.py measures the kernels."""
    A = sparse.csr_matrix(P)
    n = A.shape[0]
    world = len(bounds) - 1
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    owner_of = (np.searchsorted(bounds, pos, side='right') - 1).astype(np.int64)        # owner of every vertex
    rowlen = np.diff(A.indptr)
    row_owner = np.repeat(owner_of, rowlen)
    col_owner = owner_of[A.indices]
    entries = np.bincount(owner_of, weights=rowlen, minlength=world).astype(np.int64)
    rows = np.bincount(owner_of, minlength=world).astype(np.int64)
    remote = row_owner != col_owner
    # distinct (importing rank, imported row) pairs, then per (importer, owner)
    key = np.unique(row_owner[remote] * n + A.indices[remote])
    imp, col = key // n, key % n
    halo = np.bincount(imp, minlength=world).astype(np.int64)
    pair = np.bincount(imp * world + owner_of[col], minlength=world * world).reshape(world, world)
    peer_max = np.maximum(pair.max(axis=1), pair.max(axis=0))           # largest message a rank receives or sends
    any_halo = int(halo.sum()) > 0
    t = 2.5 + 7.7e-6 * entries + (6.6 + peer_max * 128.0 / 50e3 if any_halo else 0.0)
    return dict(entries=entries, rows=rows, halo_rows=halo, peer_max=peer_max, est_us=float(np.max(t)),
                imbalance=float(entries.max() / max(1.0, entries.mean())), crossing=int(remote.sum()))


def quotient_partition(P, order, world, segments=None, eps_levels=(0.03, 0.08, 0.15, 0.3, 0.6), seed=0):
    """A third way to give the vertices to `world` ranks, beside contiguous cuts of `order` (cut_bounds) and equal blocks: cut
    `order` into a few hundred SEGMENTS at its least-crossed positions (on a kNN graph of clustered data: cells of feature space),
    form the quotient graph of the segments (node weight = the work of a segment's rows, edge weight = stored entries between two
    segments) and distribute the NODES over the ranks -- any subset, not only runs of consecutive ones -- under a balance bound,
    by greedy growth from spread-out seeds followed by Kernighan-Lin / Fiduccia-Mattheyses style refinement (single moves and
    pair swaps of best gain).  For a growing balance allowance the assignment of least predicted sweep time (partition_cost)
    wins.  Ten clusters over eight ranks: contiguous cuts must hand two whole clusters to two ranks (imbalance 1.6); this can
    split clusters into cells and deal the cells out.
    Returns (order2, bounds, info): `order2` lists every rank's segments one after another (in the order of `order`, so the
    locality inside a rank is kept), `bounds` the rank boundaries in it -- the (order, bounds) pair RankPlan takes.  Iterates do
    not depend on the assignment: a row's entries keep their order whoever owns the row."""
    A = sparse.csr_matrix(P)
    n = A.shape[0]
    order = np.asarray(order, dtype=np.int64)
    if world <= 1:
        return order, np.array([0, n], dtype=np.int64), dict(segments=1, eps=0.0)
    S = int(segments) if segments else int(min(256, max(8 * world, 32)))
    S = max(world, min(S, n))
    cross = crossing_counts(A, order)
    edges = np.linspace(0, n, S + 1).astype(np.int64)
    cuts = [0]
    for a, b in zip(edges[1:-1], edges[2:]):            # one boundary per window: its least-crossed position
        lo = int((edges[np.searchsorted(edges, a) - 1] + a) // 2) if a > 0 else 0
        lo = max(lo, cuts[-1] + 1)
        hi = int((a + b) // 2)
        if hi <= lo:
            continue
        cuts.append(lo + int(np.argmin(cross[lo:hi])))
    segb = np.array(sorted(set(cuts)) + [n], dtype=np.int64)
    S = len(segb) - 1
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    seg_of = (np.searchsorted(segb, pos, side='right') - 1).astype(np.int64)
    rowlen = np.diff(A.indptr)
    wnode = np.bincount(seg_of, weights=rowlen + 3.0, minlength=S)
    rs, cs = np.repeat(seg_of, rowlen), seg_of[A.indices]
    Q = np.bincount(rs * S + cs, minlength=S * S).reshape(S, S).astype(np.float64)
    Q = Q + Q.T
    np.fill_diagonal(Q, 0.0)
    total = float(wnode.sum())
    rng = np.random.default_rng(seed)

    def refine(part, cap):
        load = np.bincount(part, weights=wnode, minlength=world)
        conn = np.zeros((S, world))
        for r in range(world):
            conn[:, r] = Q[:, part == r].sum(axis=1)
        for _ in range(8 * S):
            own = conn[np.arange(S), part]
            gain = conn - own[:, None]                                    # gain[v, r]: cut removed by moving v to r
            feas = (load[None, :] + wnode[:, None] <= cap)
            feas[np.arange(S), part] = False
            g = np.where(feas, gain, -np.inf)
            v, r = np.unravel_index(int(np.argmax(g)), g.shape)
            best_move = g[v, r]
            # pair swaps of the most promising candidates (keeps the loads nearly unchanged)
            best_swap, sw = 0.0, None
            cand = np.argsort(-(gain.max(axis=1)))[:24]
            for a in cand:
                for b in cand:
                    if part[a] == part[b]:
                        continue
                    ga = gain[a, part[b]] + gain[b, part[a]] - 2.0 * Q[a, b]
                    if ga > best_swap + 1e-9:
                        la = load[part[a]] - wnode[a] + wnode[b]
                        lb = load[part[b]] - wnode[b] + wnode[a]
                        if la <= cap and lb <= cap:
                            best_swap, sw = ga, (a, b)
            if best_move <= 1e-9 and sw is None:
                break

            def move(v, r):
                o = part[v]
                conn[:, o] -= Q[:, v]
                conn[:, r] += Q[:, v]
                load[o] -= wnode[v]
                load[r] += wnode[v]
                part[v] = r
            if sw is not None and best_swap > best_move:
                a, b = sw
                pa, pb = part[a], part[b]
                move(a, pb)
                move(b, pa)
            else:
                move(int(v), int(r))
        return part

    def grow(cap):
        # seeds: the heaviest node, then repeatedly the node least connected to the seeds so far; the parts take turns (lightest
        # first) adopting the unassigned node they are most connected to
        part = np.full(S, -1, dtype=np.int64)
        seeds = [int(np.argmax(wnode))]
        while len(seeds) < world:
            c = Q[:, seeds].sum(axis=1) - 1e-3 * wnode / total
            c[seeds] = np.inf
            seeds.append(int(np.argmin(c)))
        load = np.zeros(world)
        for r, v in enumerate(seeds):
            part[v] = r
            load[r] = wnode[v]
        conn = np.stack([Q[:, s] for s in seeds], axis=1)
        left = S - world
        while left > 0:
            r = int(np.argmin(load))
            free = part < 0
            score = np.where(free, conn[:, r] + 1e-9 * rng.random(S), -np.inf)
            v = int(np.argmax(score))
            if conn[v, r] <= 0:                       # nothing adjacent left: the free node nearest in the order to the part's nodes
                mine = np.flatnonzero(part == r)
                fr = np.flatnonzero(free)
                v = int(fr[np.argmin(np.min(np.abs(fr[:, None] - mine[None, :]), axis=1))])
            part[v] = r
            load[r] += wnode[v]
            conn[:, r] += Q[:, v]
            left -= 1
        return part

    def contiguous(cap):
        # consecutive runs of segments, filled up to the mean load: the starting point that already is a cut_bounds-like answer
        part = np.zeros(S, dtype=np.int64)
        acc, r = 0.0, 0
        for v in range(S):
            if acc + 0.5 * wnode[v] > total / world * (r + 1) and r < world - 1:
                r += 1
            part[v] = r
            acc += wnode[v]
        return part

    best = None
    for eps in eps_levels:
        cap = (1.0 + eps) * total / world
        if cap < wnode.max():
            continue
        for start in (grow, contiguous):
            part = refine(start(cap).copy(), cap)
            if np.bincount(part, weights=wnode, minlength=world).max() > cap * (1 + 1e-9) or len(np.unique(part)) < world:
                continue
            seg_order = np.argsort(part, kind='stable')                       # by rank, segments ascending inside a rank
            order2 = np.concatenate([order[segb[v]:segb[v + 1]] for v in seg_order])
            sizes = np.bincount(part, weights=np.diff(segb), minlength=world).astype(np.int64)
            bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            cost = partition_cost(A, order2, bounds)
            if best is None or cost['est_us'] < best[0] - 1e-9:
                best = (cost['est_us'], order2, bounds, dict(segments=S, eps=float(eps), start=start.__name__, cut_entries=cost['crossing'],
                                                              est_us=cost['est_us'], imbalance=cost['imbalance']))
    if best is None:
        b = block_bounds(n, world)
        return order, b, dict(segments=S, eps=None, start='equal blocks', est_us=partition_cost(A, order, b)['est_us'])
    return best[1], best[2], best[3]


PARTITIONS = ('auto', 'cut', 'even', 'cells')


def plan_partition(P, order, world, how='auto', dist=None, group=None):
    """(order, bounds, info) for `how` in 'cut' (contiguous blocks between the graph's pieces), 'even' (equal blocks), 'cells'
    (quotient_partition) or 'auto' (the one of cut / cells with the smaller estimated sweep time).  Every rank plans for itself from
    the same inputs; with `dist` (an initialised torch.distributed) the ranks then compare a digest of what they arrived at and
    ALL raise if they differ -- 'cells' / 'auto' go through floating-point gains and argsort ties, and two ranks on different
    numpy builds exchanging rows by different plans would hang the collectives or corrupt the halo silently."""
    if how not in PARTITIONS:
        raise ValueError('partition must be one of %s, got %r' % (', '.join(repr(p) for p in PARTITIONS), how))
    out = _plan_partition(P, order, world, how)
    if dist is not None and world > 1:
        _agree(dist, group, [np.asarray(out[0], dtype=np.int64), np.asarray(out[1], dtype=np.int64)], 'vertex partition (%s)' % how)
    return out


def _agree(dist, group, arrays, what):
    """Collective: raises RuntimeError on every rank unless all ranks hold the same arrays."""
    import hashlib
    import torch
    h = hashlib.blake2b(digest_size=7)
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    v = int.from_bytes(h.digest(), 'little')                  # < 2^56: exact in int64, and in its negation
    dev = 'cuda' if str(dist.get_backend(group)) == 'nccl' else 'cpu'
    t = torch.tensor([v, -v], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(t[0].item()), -int(t[1].item())
    if hi != lo:
        raise RuntimeError('the ranks of the job computed different %s plans (digest %x on this rank, %x .. %x over the job): '
                           'the exchange lists would not match' % (what, v, lo, hi))


def _plan_partition(P, order, world, how):
    n = sparse.csr_matrix(P).shape[0]
    if how == 'even':
        b = block_bounds(n, world)
        return order, b, dict(partition='even', **{k: v for k, v in partition_cost(P, order, b).items() if k in ('est_us', 'imbalance')})
    cut = cut_bounds(P, order, world)
    c_cut = partition_cost(P, order, cut)
    if how == 'cut' or world <= 1:
        return order, cut, dict(partition='cut', est_us=c_cut['est_us'], imbalance=c_cut['imbalance'])
    o2, b2, info = quotient_partition(P, order, world)
    if how == 'cells' or info['est_us'] < c_cut['est_us']:
        return o2, b2, dict(partition='cells', **info)
    return order, cut, dict(partition='cut', est_us=c_cut['est_us'], imbalance=c_cut['imbalance'], cells_est_us=info['est_us'])


class RankPlan:
    """What rank `rank` needs: its rows of P with columns renumbered [owned | halo] and the
    send / receive lists of the per-sweep exchange."""

    def __init__(self, P, order, bounds, rank, boundary_first=True):
        P = sparse.csr_matrix(P)
        n = P.shape[0]
        world = len(bounds) - 1
        self.rank, self.world, self.n_global = rank, world, n
        pos = np.empty(n, dtype=np.int64)          # old id -> position in the locality order
        pos[order] = np.arange(n)
        owner_of_pos = np.searchsorted(bounds, np.arange(n), side='right') - 1
        self.own = order[bounds[rank]:bounds[rank + 1]]     # global ids this rank owns, in local order
        self.n_own = len(self.own)

        def halo_of(r):
            rows = order[bounds[r]:bounds[r + 1]]
            cols = np.unique(P[rows, :].indices) if len(rows) else np.zeros(0, dtype=np.int64)
            cpos = pos[cols]
            remote = cpos[(cpos < bounds[r]) | (cpos >= bounds[r + 1])]
            remote.sort()                                    # grouped by owner (blocks are contiguous), then by position
            return remote

        my_halo_pos = halo_of(rank)
        self.halo = order[my_halo_pos]                       # global ids, in halo order
        self.n_halo = len(self.halo)
        halo_owner = owner_of_pos[my_halo_pos]
        self.recv_counts = [int(np.sum(halo_owner == r)) for r in range(world)]
        # what every peer needs from me, in the order it expects it
        send = []
        self.send_counts = []
        for dst in range(world):
            if dst == rank:
                self.send_counts.append(0)
                continue
            hp = halo_of(dst)
            mine = hp[(hp >= bounds[rank]) & (hp < bounds[rank + 1])]
            send.append(mine - bounds[rank])                 # local row indices
            self.send_counts.append(len(mine))
        self.send_idx = np.concatenate(send) if send else np.zeros(0, dtype=np.int64)
        # does ANY rank import anything?  (every rank derives the same answer from the same inputs;
        # blocks that follow the connected pieces of the graph have no halo and need no exchange)
        self.global_halo = self.n_halo + sum(len(halo_of(r)) for r in range(world) if r != rank)
        # local row order: BOUNDARY rows (needed by some peer) first, interior rows after -- the
        # boundary part of a sweep runs first, its records leave while the interior part computes
        is_b = np.zeros(self.n_own, dtype=bool)
        is_b[self.send_idx] = True
        if not boundary_first:
            is_b[:] = True
        new_of_old = np.empty(self.n_own, dtype=np.int64)
        perm_local = np.concatenate([np.flatnonzero(is_b), np.flatnonzero(~is_b)])
        new_of_old[perm_local] = np.arange(self.n_own)
        self.own = self.own[perm_local]
        self.send_idx = new_of_old[self.send_idx]
        self.n_boundary = int(is_b.sum())
        # local operator: rows = own, columns -> [0, n_own) owned, n_own + halo index otherwise
        local_of = np.full(n, -1, dtype=np.int64)
        local_of[self.own] = np.arange(self.n_own)
        local_of[self.halo] = self.n_own + np.arange(self.n_halo)
        sub = P[self.own, :]                                 # row slicing keeps the entry order of each row
        sub = sparse.csr_matrix(sub)
        cols = local_of[sub.indices]
        assert np.all(cols >= 0)
        self.P_local = sparse.csr_matrix((sub.data, cols.astype(np.int32), sub.indptr),
                                         shape=(self.n_own, self.n_own + self.n_halo))
        self.P_local.has_sorted_indices = False              # keep scipy from reordering the entries


class GatherPlan:
    """The plan of the ALL-GATHER form of the exchange (SURVEY 8e's fallback; glx.h GLX_DIST_FORM_GATHER): when a rank's halo is
    about all the rows it does not own -- the kNN graph of data without cluster structure is an expander -- there is nothing to
    select: every rank's whole block goes to every rank.  The state of a rank is `world` blocks of `cap` records in rank order
    (cap = the largest block; shorter blocks end in zero records), its own rows in block `rank`; the local operator's columns are
    numbered owner * cap + (position in the owner's block); the exchange is one in-place ncclAllGather -- no send lists, no pack
    kernel, no send buffer.  The entry order inside a row is untouched: iterates stay bit-identical."""
    gather = True

    def __init__(self, P, order, bounds, rank):
        P = sparse.csr_matrix(P)
        n = P.shape[0]
        world = len(bounds) - 1
        self.rank, self.world, self.n_global = rank, world, n
        sizes = np.diff(np.asarray(bounds, dtype=np.int64))
        self.cap = int(sizes.max()) if world else 0
        self.own = np.asarray(order[bounds[rank]:bounds[rank + 1]])
        self.n_own = len(self.own)
        self.n_boundary = self.n_own                         # every row is read by the peers
        self.own_off = rank * self.cap
        self.n_halo = (world - 1) * self.cap
        self.global_halo = world * self.n_halo
        slot = np.empty(n, dtype=np.int64)                   # global id -> its record in every rank's state
        for r in range(world):
            ids = order[bounds[r]:bounds[r + 1]]
            slot[ids] = r * self.cap + np.arange(len(ids))
        sub = sparse.csr_matrix(P[self.own, :])              # row slicing keeps the entry order of each row
        self.P_local = sparse.csr_matrix((sub.data, slot[sub.indices].astype(np.int32), sub.indptr), shape=(self.n_own, world * self.cap))
        self.P_local.has_sorted_indices = False
        self.send_counts = [0] * world
        self.recv_counts = [0] * world
        self.send_idx = np.zeros(0, dtype=np.int64)


def halo_share(P, order, bounds):
    """Sum over the ranks of the rows they import, as a share of the rows they do not own: 1 = every rank needs everybody's rows."""
    P = sparse.csr_matrix(P)
    n = P.shape[0]
    world = len(bounds) - 1
    if world <= 1:
        return 0.0
    pos = np.empty(n, dtype=np.int64)
    pos[order] = np.arange(n)
    halo = 0
    for r in range(world):
        rows = order[bounds[r]:bounds[r + 1]]
        cpos = pos[np.unique(P[rows, :].indices)] if len(rows) else np.zeros(0, dtype=np.int64)
        halo += int(np.sum((cpos < bounds[r]) | (cpos >= bounds[r + 1])))
    return halo / float((world - 1) * n)


# From this share of the foreign rows imported on, the exchange is the all-gather of whole blocks.  In the bandwidth model of
# scripts/scale_model.py (one block per link against the largest per-peer message of the halo lists) whole blocks only win when the
# ranks import nearly everything; below that the lists move fewer bytes (profiles/r05_scale_model_connected.json: share 0.5-0.7 at
# N = 2 .. 8 on the connected workload, lists 150-210 us per sweep, whole blocks 200-310 us).
GATHER_SHARE = 0.9


def make_plan(P, order, bounds, rank, exchange='auto'):
    """RankPlan (selected halo rows, all-to-all-v) or GatherPlan (whole blocks, all-gather) -- `exchange` in 'halo', 'gather',
    'auto' (the all-gather from GATHER_SHARE of the foreign rows imported on: same on every rank, it only depends on the plan)."""
    if exchange not in ('auto', 'halo', 'gather'):
        raise ValueError("exchange must be 'auto', 'halo' or 'gather', got %r" % (exchange,))
    if exchange == 'gather' or (exchange == 'auto' and len(bounds) > 2 and halo_share(P, order, bounds) >= GATHER_SHARE):
        return GatherPlan(P, order, bounds, rank)
    return RankPlan(P, order, bounds, rank)


class HipOps:
    """Rank-local sweep through libglx on torch CUDA tensors (device-pointer C-ABI)."""
    supports_graph = True

    def __init__(self, plan, C, device=None, dtype=np.float64):
        import torch
        from . import _hip
        self.torch, self._hip = torch, _hip
        self.C = C
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        device = _hip.default_device() if device is None else int(device)
        self.device = torch.device('cuda', device)
        self.devidx = device
        _assert_single_hip_runtime()
        self.lay = _hip.record_layout(C, self.dtype, True)
        self.ld = self.lay['ld']
        # two operators over the same local vector: boundary rows [0, nb) and interior rows [nb, n_own)
        nb = plan.n_boundary
        self.parts = []
        for lo, hi in ((0, nb), (nb, plan.n_own)):
            if hi <= lo:
                self.parts.append(None)
                continue
            sub = plan.P_local[lo:hi, :]
            g = _hip.DeviceGraph(sub, dtype=self.dtype, device=device, shape=(hi - lo, plan.P_local.shape[1]),
                                 keep_order=True)   # records stay in the partition's order
            n = _hip.C.c_int64(0)
            _hip.check(_hip.load().glx_graph_slots(g._h, C, 1, _hip.C.byref(n)), 'glx_graph_slots')
            self.parts.append(dict(graph=g, lo=lo, hi=hi, flags=torch.zeros(max(n.value, 1), dtype=torch.uint8, device=self.device),
                                   err=torch.zeros(64, dtype=torch.int64, device=self.device)))
        self.rec_bytes = self.lay['rec_bytes']
        self.plan = plan
        # GatherPlan: the state is `world` blocks of `cap` records and the rank's rows are block `rank` of it -- a sweep WRITES there
        # (bias, degrees, stop values stay indexed by local row); RankPlan: the own rows lead the state, offset 0
        self.own_off = int(getattr(plan, 'own_off', 0))

    def _stream(self):
        return self._hip._vp(self.torch.cuda.current_stream(self.device).cuda_stream)

    def new_state(self, rows):
        return self.torch.zeros((rows, self.ld), dtype=self.tdtype, device=self.device)

    def to_device(self, a, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def pack(self, dense, w, rows):
        """(rows, C) dense [or None = zeros] + optional (rows,) fp64 stop values -> records."""
        rec = self.new_state(rows)
        lib = self._hip.load()
        d = None if dense is None else self.to_device(dense, self.tdtype)
        wt = None if w is None else self.to_device(w, self.torch.float64)
        self._hip.check(lib.glx_pack_records_dev(None if d is None else d.data_ptr(), rec.data_ptr(), rows, self.C,
                                                  self._hip._dt(self.dtype), 1, None if wt is None else wt.data_ptr(),
                                                  self._stream()), 'glx_pack_records_dev')
        self.torch.cuda.current_stream(self.device).synchronize()   # d / wt die here
        return rec

    def unpack(self, rec, rows):
        out = self.torch.empty((rows, self.C), dtype=self.tdtype, device=self.device)
        self._hip.check(self._hip.load().glx_unpack_records_dev(rec.data_ptr(), out.data_ptr(), rows, self.C,
                                                                 self._hip._dt(self.dtype), 1, self._stream()),
                        'glx_unpack_records_dev')
        return out.cpu().numpy()

    def set_bias(self, bias_rec):
        self.bias = bias_rec
        for part in self.parts:
            if part is not None:
                self._hip.check(self._hip.load().glx_bias_flags_dev(
                    part['graph']._h, self.C, 1, bias_rec.data_ptr() + part['lo'] * self.rec_bytes, part['flags'].data_ptr(),
                    self._stream()), 'glx_bias_flags_dev')

    def set_stop_vectors(self, deg, vinf):
        self.deg = self.to_device(deg, self.torch.float64)
        self.vinf = self.to_device(vinf, self.torch.float64)

    def sweep_part(self, which, xin, xout, want_err):
        """Rows of part `which` (0 = boundary, 1 = interior): xout[rows] = bias[rows] + P[rows] xin."""
        part = self.parts[which]
        if part is None:
            return None
        lo = part['lo']
        if want_err:
            part['err'].zero_()
        self._hip.check(self._hip.load().glx_sweep_step_dev(
            part['graph']._h, self.C, 1, xin.data_ptr(), xout.data_ptr() + (self.own_off + lo) * self.rec_bytes,
            self.bias.data_ptr() + lo * self.rec_bytes, part['flags'].data_ptr(), self.deg.data_ptr() + lo * 8,
            self.vinf.data_ptr() + lo * 8, part['err'].data_ptr() if want_err else None, self._stream()), 'glx_sweep_step_dev')
        if want_err:   # fp64 bit patterns of non-negative values order like the values
            return part['err'].max().view(1).view(self.torch.float64)
        return None

    def sweep(self, xin, xout, want_err):
        e0 = self.sweep_part(0, xin, xout, want_err)
        e1 = self.sweep_part(1, xin, xout, want_err)
        return combine_err(self.torch, e0, e1)

    def index_rows(self, rec, idx):
        return rec.index_select(0, idx)

    def close(self):
        for part in self.parts:
            if part is not None:
                part['graph'].close()


def combine_err(torch, e0, e1):
    if e0 is None:
        return e1
    if e1 is None:
        return e0
    return torch.maximum(e0, e1)


def _assert_single_hip_runtime():
    """torch bundles its own libamdhip64; libglx must have bound to THAT copy (it does when
    torch is imported before libglx is first loaded), otherwise torch's pointers and streams
    would belong to a different HIP runtime instance."""
    paths = set()
    try:
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    paths.add(line.split()[-1])
    except OSError:
        return
    if len(paths) > 1:
        raise RuntimeError('two HIP runtimes are loaded (%s): import torch before the first graphlearning_amd GPU call '
                           'in multi-GPU processes' % sorted(paths))


class DistSweep:
    """The distributed Poisson sweep.  `ops` provides the rank-local kernel, `dist` is
    torch.distributed (already initialised), tensors live wherever `ops` puts them."""

    def __init__(self, plan, ops, dist, group=None):
        import torch
        self.torch = torch
        self.plan, self.ops, self.dist, self.group = plan, ops, dist, group
        self.n_loc = plan.n_own + plan.n_halo if not getattr(plan, 'gather', False) else plan.world * plan.cap
        self.own_off = int(getattr(plan, 'own_off', 0))     # GatherPlan: the rank's rows sit in block `rank` of the state
        self.send_idx = ops.to_device(plan.send_idx.astype(np.int64))
        self.in_splits = list(plan.send_counts)
        self.out_splits = list(plan.recv_counts)
        self.exchanges = 0
        # a backend without device collectives (gloo) with device-resident records: stage through the host
        self._stage_host = bool(getattr(ops, 'supports_graph', False)) and dist.get_backend(group) == 'gloo'
        self._force_coll = _force_collectives()   # test hook: collectives at world 1

    def exchange(self, x, async_op=False):
        """Boundary records of x[0:n_own] -> the peers' halo regions x[n_own:].  With async_op the
        collective is only enqueued; the returned handle's wait() orders later work behind it."""
        p = self.plan
        if (p.world == 1 or p.global_halo == 0) and not self._force_coll:
            return None
        if getattr(p, 'gather', False):
            # whole blocks to everybody (GatherPlan): an all-gather straight into the blocks of the state (the scipy stand-in of the
            # tests and host-staged records; the library's own engine does it in place on the device, csrc/dist.hip)
            self.exchanges += 1
            mine = x[self.own_off:self.own_off + p.cap].clone()
            blocks = [x[r * p.cap:(r + 1) * p.cap] for r in range(p.world)]
            if self._stage_host:
                host = [self.torch.empty(b.shape, dtype=b.dtype) for b in blocks]
                self.dist.all_gather(host, mine.cpu(), group=self.group)
                for b, h in zip(blocks, host):
                    b.copy_(h)
            else:
                self.dist.all_gather(blocks, mine, group=self.group)
            return None
        send = self.ops.index_rows(x, self.send_idx)
        recv = x[p.n_own:]
        self.exchanges += 1
        if self._stage_host:
            send_h = send.cpu()
            recv_h = self.torch.empty(recv.shape, dtype=recv.dtype)
            self.dist.all_to_all_single(recv_h, send_h, output_split_sizes=self.out_splits, input_split_sizes=self.in_splits,
                                        group=self.group)
            recv.copy_(recv_h)
            return None
        work = self.dist.all_to_all_single(recv, send, output_split_sizes=self.out_splits, input_split_sizes=self.in_splits,
                                           group=self.group, async_op=async_op)
        return work if async_op else None

    def step(self, xin, xout, want_err):
        """One distributed sweep: boundary rows, start their exchange, interior rows meanwhile, join."""
        ops = self.ops
        if hasattr(ops, 'sweep_part'):
            e0 = ops.sweep_part(0, xin, xout, want_err)
            work = self.exchange(xout, async_op=True)
            e1 = ops.sweep_part(1, xin, xout, want_err)
            if work is not None:
                work.wait()
            return combine_err(self.torch, e0, e1)
        e = ops.sweep(xin, xout, want_err)
        self.exchange(xout)
        return e

    def _all_reduce_max(self, e):
        if self._stage_host:
            h = e.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group)
            e.copy_(h)
        else:
            self.dist.all_reduce(e, op=self.dist.ReduceOp.MAX, group=self.group)

    def setup(self, Db_own, w0_own, deg_own, vinf_own):
        """Rank-local problem data (rows in this rank's local order)."""
        ops, p = self.ops, self.plan
        ops.set_bias(ops.pack(Db_own, None, p.n_own))
        ops.set_stop_vectors(deg_own, vinf_own)
        self.init_rec = ops.pack(None, np.ascontiguousarray(w0_own, dtype=np.float64), p.n_own)   # u = 0 (ssl.py:645), w = w0
        self.xa = ops.new_state(self.n_loc)
        self.xb = ops.new_state(self.n_loc)
        self.cur = 0
        self._graph = None
        self._graph_err = None
        self._graph_ok = bool(getattr(ops, 'supports_graph', False)) and not self._stage_host

    def close(self):
        """Release the captured device graph and the rank-local state while the process group is
        still alive (not at interpreter exit)."""
        if getattr(self, '_graph', None) is not None:
            try:
                import torch
                torch.cuda.synchronize()
            except Exception:
                pass
            self._graph = None
            self._graph_err = None
        self.xa = self.xb = self.init_rec = None

    def reset(self):
        """Owned rows <- initial records; halo filled by one exchange.  Allocation-free apart from
        the exchange's send buffer; without an exchange it can be stream-captured."""
        p = self.plan
        self.xa.zero_()
        self.xa[self.own_off:self.own_off + p.n_own].copy_(self.init_rec)
        self.exchange(self.xa)
        self.cur = 0

    def _head(self, nsweeps, min_iter):
        """The first `nsweeps` sweeps, which the stop test cannot cut short; returns the device
        scalar holding max|v - vinf| after the last one when the stop test will need it."""
        bufs = [self.xa, self.xb]
        e = None
        for T in range(nsweeps):
            want = (T + 1) >= min_iter
            e = self.step(bufs[self.cur], bufs[self.cur ^ 1], want)
            self.cur ^= 1
        return e

    def _exchanges(self):
        """Does a sweep contain a collective (halo exchange)?"""
        p = self.plan
        return (p.world > 1 and p.global_halo > 0) or self._force_coll

    def run(self, min_iter, max_iter, err0=None):
        """All sweeps; returns T.  Stop test (ssl.py:667): T = first T >= min_iter with
        max over ALL vertices |deg w_T - vinf| <= 1/n_global.  When the sweeps are purely local (no
        halo anywhere, or one rank) the min_iter sweeps that always run are captured once into a
        device graph and replayed.  Collectives are NEVER captured: the process group's watchdog
        thread polls the completion events of the work it tracks, and an event recorded inside a
        capture makes that query fail (hipErrorCapturedEvent) and the watchdog abort the process --
        seen in about 1 of 20 single-rank runs when the exchanges were part of the graph.  With a
        halo the sequence [boundary rows | exchange | interior rows] therefore runs eagerly."""
        torch, dist, ops, p = self.torch, self.dist, self.ops, self.plan
        thresh = 1.0 / p.n_global
        head = min(min_iter, max_iter)
        e = None
        if self._graph_ok and head > 0 and not self._exchanges():
            if self._graph is None:
                try:
                    self.reset()                                   # eager warm-up
                    self._head(1, min_iter + 2)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):
                        self.reset()
                        self._graph_err = self._head(head, min_iter)
                    self._graph = g
                except Exception as exc:
                    self._graph_ok = False
                    self._graph = None
                    torch.cuda.synchronize()
                    import warnings
                    warnings.warn('DistSweep: device-graph capture unavailable (%s); running eagerly' % (exc,))
            if self._graph is not None:
                self._graph.replay()
                self.cur = head & 1
                e = self._graph_err
        if self._graph is None or head == 0:
            self.reset()
            e = self._head(head, min_iter)
        if e is not None and head >= min_iter and (p.world > 1 or self._force_coll):
            self._all_reduce_max(e)                                # stop test of the last head sweep, outside any capture
        bufs = [self.xa, self.xb]
        T = head
        err_T = err0 if head == 0 else (float(e.item()) if e is not None else None)
        while T < max_iter:
            if T >= min_iter:
                if err_T is None:
                    raise RuntimeError('stop test needs the error of v_%d' % T)
                if not (err_T > thresh):
                    break
            xin, xout = bufs[self.cur], bufs[self.cur ^ 1]
            want = (T + 1) >= min_iter
            e = self.step(xin, xout, want)
            if want:
                if p.world > 1 or self._force_coll:
                    self._all_reduce_max(e)
                err_T = float(e.item())
            self.cur ^= 1
            T += 1
        return T

    def result_own(self):
        x = [self.xa, self.xb][self.cur]
        return self.ops.unpack(x[self.own_off:self.own_off + self.plan.n_own], self.plan.n_own)


# ---- libglx-owned communicator and sweep (include/glx.h: glx_dist_*) ----------------------------------------------
def init_comm(dist, device=None, group=None):
    """A libglx RCCL communicator over the ranks of `dist` (any initialised torch.distributed backend serves as the
    bootstrap channel: rank 0's 128-byte unique id is broadcast as a Python object).  One rank: no transport."""
    from . import _hip
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1 and not _force_collectives():
        return _hip.Comm(1, 0, None, device)
    uid = [_hip.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0, group=group)
    return _hip.Comm(world, rank, uid[0], device)


# Test hook (bench.py --force-collectives, tests/test_gpu_dist.py): the collectives of the exchange and of the stop test are issued
# even by a one-rank job, so that the code path of a real multi-rank run executes on the one GPU a test box has.
FORCE_COLLECTIVES = False


def _force_collectives():
    return bool(FORCE_COLLECTIVES)


def glx_dist_sweep(comm, plan, C, dtype=np.float64, force_exchange=False, use_hipgraph=True, form='auto'):
    """The rank's glx_dist_sweep for a RankPlan (or ShardPlan).  If ANY rank imports a halo, every rank takes part in the
    per-sweep exchange (with nothing to send or receive where it has no halo): the collective calls around it -- the capture
    self-test's verdict, the stop test's all-reduce -- must be issued by all ranks alike."""
    from . import _hip
    any_halo = int(getattr(plan, 'global_halo', 0)) > 0 and int(getattr(plan, 'world', 1)) > 1
    return _hip.DistSweep(comm, plan.P_local, plan.n_boundary, plan.send_counts, plan.send_idx, plan.recv_counts, plan.n_global, C,
                          dtype=dtype, force_exchange=bool(force_exchange or any_halo), use_hipgraph=use_hipgraph, form=form,
                          gather_cap=plan.cap if getattr(plan, 'gather', False) else None)


def run_stepwise(ds, plan, dist, min_iter, max_iter, err0=None, group=None):
    """The distributed sweep with libglx doing every rank-local piece (boundary rows, pack, interior rows, local
    maxima) and `dist` -- any backend, host tensors -- moving the packed records: how several ranks sharing ONE GPU
    (tests) or a machine without RCCL run the glx_dist_sweep object.  Returns T."""
    import torch
    world = dist.get_world_size(group)

    def exchange(next_iterate):
        if plan.global_halo == 0:
            return
        if getattr(plan, 'gather', False):       # whole blocks: the all-gather of the library's form, here through `dist`
            mine = torch.from_numpy(np.ascontiguousarray(ds.get_send()))
            blocks = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(blocks, mine, group=group)
            rank = dist.get_rank(group)
            ds.put_halo(np.concatenate([b.numpy() for r, b in enumerate(blocks) if r != rank], axis=0), next_iterate)
            return
        send = torch.from_numpy(np.ascontiguousarray(ds.get_send()))
        recv = torch.empty((plan.n_halo, ds.lay['ld']), dtype=send.dtype)
        dist.all_to_all_single(recv, send, output_split_sizes=list(plan.recv_counts), input_split_sizes=list(plan.send_counts),
                               group=group)
        ds.put_halo(recv.numpy(), next_iterate)

    ds.begin()
    exchange(False)
    thresh = 1.0 / plan.n_global
    T, err_T = 0, err0
    while T < max_iter:
        if T >= min_iter and not (err_T > thresh):
            break
        want = (T + 1) >= min_iter
        ds.boundary(want)
        exchange(True)
        e = ds.interior(want)
        if want:
            t = torch.tensor([e], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            err_T = float(t.item())
        T += 1
    return T


def poisson_fit_glx(W, train_ind, train_labels, dist, comm=None, device=None, min_iter=50, max_iter=1000, order=None, group=None,
                    gather=True, partition='cut', dtype=np.float64, check_every=8, stepwise=False, force_exchange=False, exchange='auto'):
    """ssl.poisson(solver='gradient_descent').fit across the ranks of `dist` with the library-owned sweep: planner on
    the host, then ONE collective call (glx_poisson_sweep_dist) runs every sweep, exchange and stop test on the
    device (stepwise=True: the same object with `dist` as the transport, see run_stepwise).  Returns (u, T) as
    poisson_fit_distributed does."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    prob = poisson_problem(W, train_ind, train_labels)
    P = prob['P']
    n = P.shape[0]
    if order is None:
        order = locality_order(P)
    order, bounds, _ = plan_partition(P, order, world, partition, dist, group)
    plan = make_plan(P, order, bounds, rank, exchange)     # selected halo rows (all-to-all-v) or whole blocks (all-gather)
    own_comm = comm is None and not stepwise
    if stepwise and comm is None:
        from . import _hip
        comm = _hip.Comm(world, rank, None, device)      # rank identity only: `dist` is the transport
    elif comm is None:
        comm = init_comm(dist, device, group)
    ds = glx_dist_sweep(comm, plan, prob['k'], dtype=dtype, force_exchange=force_exchange or _force_collectives())
    own = plan.own
    ds.set_problem(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
    err0 = initial_error(prob['w0'][own], prob['deg'][own], prob['vinf'][own], dist, group) if min_iter == 0 else 0.0
    if stepwise:
        T = run_stepwise(ds, plan, dist, min_iter, max_iter, err0, group)
    else:
        T, _ = ds.run(min_iter, max_iter, check_every, err0)
    u_own = ds.fetch()
    stats = ds.stats()
    ds.close()
    if own_comm or stepwise:
        comm.close()
    if not gather:
        return u_own, T, plan, stats
    parts = [None] * world
    dist.all_gather_object(parts, (own, u_own), group=group)
    u = np.zeros((n, prob['k']), dtype=u_own.dtype)
    for ids, block in parts:
        u[ids] = block
    return u, T


def knnsearch_distributed(X, k, dist, device=None, similarity='euclidean', group=None):
    """weightmatrix.knnsearch with the QUERY rows sharded over the ranks: every
    rank holds all of X, searches its contiguous block of queries on its own GPU
    (glx_knn_bruteforce_range) and the blocks are all_gathered -- the only communication."""
    import torch
    from . import _hip
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = _hip.default_device() if device is None else int(device)
    n = X.shape[0]
    bounds = block_bounds(n, world)
    ind, dst = _hip.knn_bruteforce(X, k, similarity=similarity, device=device,
                                   query_range=(int(bounds[rank]), int(bounds[rank + 1])))
    if world == 1:
        return ind, dst
    rows = int(np.max(np.diff(bounds)))               # pad to equal blocks for all_gather
    dev = torch.device('cuda', device)
    ti = torch.full((rows, k), -1, dtype=torch.int64, device=dev)
    td = torch.zeros((rows, k), dtype=torch.float64, device=dev)
    ti[:len(ind)] = torch.from_numpy(ind).to(dev)
    td[:len(dst)] = torch.from_numpy(dst).to(dev)
    gi = [torch.empty_like(ti) for _ in range(world)]
    gd = [torch.empty_like(td) for _ in range(world)]
    dist.all_gather(gi, ti, group=group)
    dist.all_gather(gd, td, group=group)
    ind_all = np.concatenate([gi[r][:bounds[r + 1] - bounds[r]].cpu().numpy() for r in range(world)])
    dst_all = np.concatenate([gd[r][:bounds[r + 1] - bounds[r]].cpu().numpy() for r in range(world)])
    return ind_all, dst_all


def initial_error(w0, deg, vinf, dist=None, group=None, torch=None):
    """max |deg*w0 - vinf| over all vertices (needed only when min_iter == 0)."""
    e = float(np.max(np.abs(deg * w0 - vinf))) if len(w0) else 0.0
    if dist is not None and dist.get_world_size(group) > 1:
        parts = [None] * dist.get_world_size(group)
        dist.all_gather_object(parts, e, group=group)     # backend-agnostic (a python float per rank)
        e = max(parts)
    return e


def poisson_problem(W, train_ind, train_labels):
    """Global host-side setup of the gradient-descent solver (reference ssl.py:615-645):
    returns P = D^-1 W^T, Db, w0 = D^-1 v0, deg, vinf, number of classes."""
    from . import graph as graph_mod
    from . import ssl as ssl_mod
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = W - sparse.spdiags(W.diagonal(), 0, n, n)
    G = graph_mod.graph(W)
    source, k = ssl_mod._poisson_source(n, np.asarray(train_ind), np.asarray(train_labels))
    D = G.degree_matrix(p=-1)
    P = sparse.csr_matrix(D * W.transpose())
    deg = G.degree_vector()
    v0 = np.zeros(n)
    v0[train_ind] = 1
    v0 = v0 / np.sum(v0)
    return dict(P=P, Db=D * source, w0=v0 / deg, deg=deg, vinf=deg / np.sum(deg), k=k)


def poisson_fit_distributed(W, train_ind, train_labels, dist, ops_factory, min_iter=50, max_iter=1000, order=None,
                            group=None, gather=True, partition='cut', exchange='halo'):
    """ssl.poisson(solver='gradient_descent').fit across the ranks of `dist`.
    Every rank holds the whole (host) graph and calls this collectively.  Returns (u, T) with
    u the full (n,C) matrix on every rank (gather=True) or this rank's rows.  partition: 'cut' =
    block boundaries that follow the graph's pieces (cut_bounds), 'even' = equal blocks."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    prob = poisson_problem(W, train_ind, train_labels)
    P = prob['P']
    n = P.shape[0]
    if order is None:
        order = locality_order(P)
    order, bounds, _ = plan_partition(P, order, world, partition, dist, group)
    plan = make_plan(P, order, bounds, rank, exchange)
    ops = ops_factory(plan, prob['k'])
    sweep = DistSweep(plan, ops, dist, group)
    own = plan.own
    sweep.setup(prob['Db'][own], prob['w0'][own], prob['deg'][own], prob['vinf'][own])
    err0 = initial_error(prob['w0'][own], prob['deg'][own], prob['vinf'][own], dist, group, torch) if min_iter == 0 else None
    T = sweep.run(min_iter, max_iter, err0)
    u_own = sweep.result_own()
    sweep.close()                      # captured graph and device state go while the group is alive
    if hasattr(ops, 'close'):
        ops.close()
    if not gather:
        return u_own, T, plan
    parts = [None] * world
    dist.all_gather_object(parts, (own, u_own), group=group)
    u = np.zeros((n, prob['k']), dtype=u_own.dtype)
    for ids, block in parts:
        u[ids] = block
    return u, T


# ---- vertex-partitioned conjugate gradient (tolerance mode) ----------------------------------------------------------
# ssl.laplace / ssl.randomwalk solve SPD systems with utils.conjgrad (reference utils.py:483-532).  Across ranks: every
# iteration exchanges the boundary rows of p (the same all-to-all-v as the sweep), multiplies the rank's rows, and adds the
# ranks' column sums of p*Ap and r*r with one all-reduce each.  The summation order then depends on the
# partition, so this is the tolerance mode (iterates within 1e-5 of the reference, identical labels: `reduce='tree'` on one
# GPU makes the same trade).  Poisson's default CG solve is singular and its 140+ iterations amplify a reordered reduction into
# another iteration count and iterates that differ by far more than 1e-5: poisson_cg_fit_distributed runs it across
# ranks under a residual contract only (same stop, labels agree away from ties); the bit-identical form stays on one GPU.
class CgScipyOps:
    """Rank-local pieces of the distributed CG on numpy arrays (CPU tests over gloo)."""
    supports_graph = False

    def __init__(self, plan, C):
        self.plan, self.C = plan, C

    def to_device(self, a, dtype=None):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a))

    def index_rows(self, x, idx):
        return x.index_select(0, idx)

    def state(self, dense_own, rows):
        import torch
        t = torch.zeros((rows, self.C), dtype=torch.float64)
        if dense_own is not None:
            t[:dense_own.shape[0]] = torch.from_numpy(np.ascontiguousarray(dense_own, dtype=np.float64))
        return t

    def spmm(self, p_loc, Ap):
        Ap.copy_(self.to_device(self.plan.P_local * p_loc.numpy()))

    def dots(self, a, b):
        n = self.plan.n_own
        return np.sum(a[:n].numpy() * b[:n].numpy(), axis=0)

    def axpy2(self, x, r, p_loc, Ap, alpha):
        n = self.plan.n_own
        x.numpy()[...] += alpha * p_loc[:n].numpy()
        r.numpy()[...] -= alpha * Ap.numpy()

    def xpby(self, p_loc, r, beta):
        n = self.plan.n_own
        p_loc.numpy()[:n] = r.numpy() + beta * p_loc[:n].numpy()

    def to_host(self, x):
        return np.array(x.numpy())

    def close(self):
        pass


class CgHipOps:
    """Rank-local pieces of the distributed CG on torch CUDA tensors in libglx's record layout: the sliced-ELL SpMM
    (glx_sweep_step_dev) and the dense vector kernels of csrc/vecops.hip (glx_rec_*_dev)."""
    supports_graph = True

    def __init__(self, plan, C, device=None, dtype=np.float64):
        import torch
        from . import _hip
        self.torch, self._hip, self.plan, self.C = torch, _hip, plan, C
        if getattr(plan, 'gather', False):
            raise ValueError('CgHipOps works on a RankPlan ([owned | halo] columns); the all-gather form (GatherPlan) is a form of the sweep only')
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        device = _hip.default_device() if device is None else int(device)
        self.device = torch.device('cuda', device)
        _assert_single_hip_runtime()
        self.lay = _hip.record_layout(C, self.dtype, False)
        self.ld = self.lay['ld']
        self.graph = _hip.DeviceGraph(plan.P_local, dtype=self.dtype, device=device, shape=plan.P_local.shape, keep_order=True)
        nscr = int(_hip.load().glx_rec_dots_scratch(max(plan.n_own, 1), C))
        self.scratch = torch.zeros(nscr, dtype=torch.float64, device=self.device)
        self.dot_out = torch.zeros(max(C, 1), dtype=torch.float64, device=self.device)

    def _stream(self):
        return self._hip._vp(self.torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, a, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def index_rows(self, x, idx):
        return x.index_select(0, idx)

    def state(self, dense_own, rows):
        rec = self.torch.zeros((rows, self.ld), dtype=self.tdtype, device=self.device)
        if dense_own is not None:
            d = self.to_device(dense_own, self.tdtype)
            self._hip.check(self._hip.load().glx_pack_records_dev(d.data_ptr(), rec.data_ptr(), dense_own.shape[0], self.C,
                                                                  self._hip._dt(self.dtype), 0, None, self._stream()), 'glx_pack_records_dev')
            self.torch.cuda.current_stream(self.device).synchronize()
        return rec

    def spmm(self, p_loc, Ap):
        self._hip.check(self._hip.load().glx_sweep_step_dev(self.graph._h, self.C, 0, p_loc.data_ptr(), Ap.data_ptr(), None, None, None, None,
                                                            None, self._stream()), 'glx_sweep_step_dev')

    def dots(self, a, b):
        self._hip.check(self._hip.load().glx_rec_dots_dev(a.data_ptr(), b.data_ptr(), self.plan.n_own, self.C, self._hip._dt(self.dtype), 0,
                                                          self.scratch.data_ptr(), self.dot_out.data_ptr(), self._stream()), 'glx_rec_dots_dev')
        return self.dot_out[:self.C].cpu().numpy().copy()

    def axpy2(self, x, r, p_loc, Ap, alpha):
        al = self.to_device(np.ascontiguousarray(alpha, dtype=np.float64))
        self._hip.check(self._hip.load().glx_rec_axpy2_dev(x.data_ptr(), r.data_ptr(), p_loc.data_ptr(), Ap.data_ptr(), al.data_ptr(),
                                                           self.plan.n_own, self.C, self._hip._dt(self.dtype), 0, self._stream()), 'glx_rec_axpy2_dev')
        self.torch.cuda.current_stream(self.device).synchronize()      # `al` dies here

    def xpby(self, p_loc, r, beta):
        be = self.to_device(np.ascontiguousarray(beta, dtype=np.float64))
        self._hip.check(self._hip.load().glx_rec_xpby_dev(p_loc.data_ptr(), r.data_ptr(), be.data_ptr(), self.plan.n_own, self.C,
                                                          self._hip._dt(self.dtype), 0, self._stream()), 'glx_rec_xpby_dev')
        self.torch.cuda.current_stream(self.device).synchronize()

    def to_host(self, x):
        out = self.torch.empty((self.plan.n_own, self.C), dtype=self.tdtype, device=self.device)
        self._hip.check(self._hip.load().glx_unpack_records_dev(x.data_ptr(), out.data_ptr(), self.plan.n_own, self.C, self._hip._dt(self.dtype), 0,
                                                                self._stream()), 'glx_unpack_records_dev')
        return out.cpu().numpy()

    def close(self):
        self.graph.close()


def cg_distributed(A, B, dist, ops_factory, tol=1e-10, max_iter=100000, order=None, partition='even', group=None, gather=True):
    """utils.conjgrad (reference utils.py:483-532: multi right-hand-side CG from x = 0, per-column alpha / beta, global stop
    sqrt(sum over all columns of |r|^2) <= tol) for a symmetric positive definite A whose rows are partitioned over the
    ranks of `dist`; every rank holds the host matrix and calls this collectively.  Per iteration: one halo exchange of the
    boundary rows of p, the rank's rows of A p, two all-reduces of C column sums.  Returns (x, iterations, err) with x the full
    (n, C) solution (gather=True) or (x_own, iterations, err, plan)."""
    import torch
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    A = sparse.csr_matrix(A)
    n = A.shape[0]
    B = np.asarray(B, dtype=np.float64)
    C = B.shape[1]
    if order is None:
        order = locality_order(A)
    order, bounds, _ = plan_partition(A, order, world, partition, dist, group)
    plan = RankPlan(A, order, bounds, rank)
    ops = ops_factory(plan, C)
    xch = DistSweep(plan, ops, dist, group)          # its exchange(): boundary records of p[0:n_own] -> the peers' halo regions p[n_own:]
    n_own, n_loc = plan.n_own, plan.n_own + plan.n_halo
    own = plan.own

    def allsum(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        if world > 1:
            if dist.get_backend(group) == 'nccl':
                t = t.to(ops.device)
            dist.all_reduce(t, group=group)
        return t.cpu().numpy()

    x = ops.state(None, n_own)
    r = ops.state(B[own], n_own)
    p = ops.state(B[own], n_loc)                      # p = r; the halo rows arrive with the first exchange
    Ap = ops.state(None, n_own)
    rsold = allsum(ops.dots(r, r))
    err, it = 1.0, 0
    while err > tol and it < max_iter:                # utils.py:519
        it += 1
        xch.exchange(p)
        ops.spmm(p, Ap)
        with np.errstate(divide='ignore', invalid='ignore'):
            alpha = rsold / allsum(ops.dots(p, Ap))
        ops.axpy2(x, r, p, Ap, alpha)
        rsnew = allsum(ops.dots(r, r))
        err = float(np.sqrt(np.sum(rsnew)))
        with np.errstate(divide='ignore', invalid='ignore'):
            ops.xpby(p, r, rsnew / rsold)
        rsold = rsnew
    x_own = ops.to_host(x)
    ops.close()
    if not gather:
        return x_own, it, err, plan
    parts = [None] * world
    dist.all_gather_object(parts, (own, x_own), group=group)
    xf = np.zeros((n, C), dtype=x_own.dtype)
    for ids, block in parts:
        xf[ids] = block
    return xf, it, err


def laplace_fit_distributed(W, train_ind, train_labels, dist, ops_factory, normalization='combinatorial', tau=0, mean_shift=False, tol=1e-5,
                            order=None, partition='even', group=None):
    """ssl.laplace(W, ...).fit across the ranks of `dist` (reference ssl.py:1206-1261; reweighting 'none', order 1): the
    Jacobi-scaled Dirichlet system M A M, A = L[unl, unl], embedded in the full vertex set -- rows and columns of the labelled
    vertices replaced by the identity, zero right-hand side there -- so that the rank partition is one of ALL vertices and x
    stays zero on the labelled rows; solved by cg_distributed.  Tolerance mode: labels identical, iterates within 1e-5 of the
    reference.  Returns (u (n, C), CG iterations)."""
    from . import graph as graph_mod
    from . import utils
    n = W.shape[0]
    G = graph_mod.graph(W)
    tau_v = np.ones(n) * tau if np.isscalar(tau) else np.asarray(tau, dtype=np.float64)
    L = sparse.csr_matrix(sparse.spdiags(tau_v, 0, n, n) + G.laplacian(normalization=normalization))
    train_ind = np.asarray(train_ind)
    k = len(np.unique(train_labels))
    F = utils.labels_to_onehot(np.asarray(train_labels), k)
    unl = np.ones(n, dtype=bool)
    unl[train_ind] = False
    b = -L[:, train_ind] * F                                    # ssl.py:1236
    Mv = 1 / np.sqrt(L.diagonal() + 1e-10)                      # ssl.py:1244-1246 (A_ii = L_ii)
    Z = sparse.spdiags(unl.astype(np.float64), 0, n, n).tocsr()
    M = sparse.spdiags(Mv, 0, n, n).tocsr()
    Afull = sparse.csr_matrix(Z * (M * L * M) * Z + sparse.spdiags((~unl).astype(np.float64), 0, n, n))
    Afull.eliminate_zeros()
    rhs = (Mv[:, None] * b) * unl[:, None]
    if order is None:
        order = locality_order(L)
    x, it, _ = cg_distributed(Afull, rhs, dist, ops_factory, tol=tol, order=order, partition=partition, group=group)
    u = Mv[:, None] * x                                         # ssl.py:1250
    u[train_ind, :] = F                                         # ssl.py:1253-1255
    if mean_shift:
        u -= np.mean(u, axis=0)
    return u, it


def poisson_cg_fit_distributed(W, train_ind, train_labels, dist, ops_factory, tol=1e-3, order=None, partition='even', group=None):
    """ssl.poisson(W) -- the DEFAULT solver, conjugate gradient -- .fit across the ranks of `dist` (reference ssl.py:608-629): the
    singular system L_normalized x = D^-1/2 source of the graph without its diagonal, by cg_distributed, u = D^-1/2 x.  The contract
    is weaker than the other distributed solvers' and said so: the stop is the reference's (sqrt of the summed squared residuals
    <= tol, default 1e-3), but 100+ iterations on a singular system turn a reordered reduction into iterates that differ from the
    one-GPU, reference-order solve by up to the order of tol and into an iteration count a few steps off -- both runs are solutions
    to the same residual bound, the labels agree except on vertices whose two largest scores are closer than that.  (On a graph with
    several components the reference's own system is inconsistent -- the source need not sum to zero per component -- and neither
    solve means anything: tests/cg_worker.py uses connected graphs.)  Use
    ssl.poisson on one GPU where the reference's exact iterates matter.  Returns (u (n, C), CG iterations)."""
    from . import graph as graph_mod
    from . import ssl as ssl_mod
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = sparse.csr_matrix(W - sparse.spdiags(W.diagonal(), 0, n, n))          # ssl.py:615-617
    G = graph_mod.graph(W)
    source, _ = ssl_mod._poisson_source(n, np.asarray(train_ind), np.asarray(train_labels))
    L = sparse.csr_matrix(G.laplacian(normalization='normalized'))
    D = G.degree_matrix(p=-0.5)
    if order is None:
        order = locality_order(L)
    x, it, _ = cg_distributed(L, D * source, dist, ops_factory, tol=tol, order=order, partition=partition, group=group)
    return D * x, it


def randomwalk_fit_distributed(W, train_ind, train_labels, dist, ops_factory, alpha=0.95, order=None, partition='even', group=None):
    """ssl.randomwalk(W, alpha).fit across the ranks of `dist` (reference ssl.py:1765-1793): the Jacobi-scaled system
    M L M with L = (1-alpha) I + alpha L_normalized, tol 1e-6, by cg_distributed.  Returns (u, CG iterations)."""
    from . import graph as graph_mod
    from . import utils
    n = W.shape[0]
    W = sparse.csr_matrix(W)
    W = W - sparse.spdiags(W.diagonal(), 0, n, n)
    G = graph_mod.graph(W)
    L = (1 - alpha) * sparse.identity(n) + alpha * G.laplacian(normalization='normalized')
    Md = sparse.spdiags(1 / np.sqrt(L.diagonal() + 1e-10), 0, n, n).tocsr()
    k = len(np.unique(train_labels))
    onehot = utils.labels_to_onehot(np.asarray(train_labels), k)
    Y = np.zeros((n, onehot.shape[1]))
    Y[np.asarray(train_ind), :] = onehot
    A = sparse.csr_matrix(Md * L * Md)
    if order is None:
        order = locality_order(A)
    x, it, _ = cg_distributed(A, Md * Y, dist, ops_factory, tol=1e-6, order=order, partition=partition, group=group)
    return Md * x, it


def ssl_trials_distributed(model, trainsets, labels, dist, tag='', save_results=True, overwrite=False, num_trials=-1, group=None):
    """ssl.ssl_trials with the training sets shared out over the ranks of `dist` -- the parallel axis
    the reference gives to joblib workers (`num_cores`, reference ssl.py:390-396).  Every rank holds
    its own model on its own GPU and runs a contiguous share of the trials; there is no data-path
    communication, the result rows are gathered once at the end and rank 0 writes the file in the
    original order (same format as ssl_trials).  As with the reference's worker processes, a learner
    with class priors warm-starts its volume weights from the previous trial OF THE SAME RANK.
    Returns the list of rows on every rank (None when the run is refused because the results file exists)."""
    import os
    from . import ssl as ssl_mod
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if num_trials > 0:
        trainsets = trainsets[:num_trials]
    # the reference aborts BEFORE any work when the results file exists (ssl.py:330-337): rank 0 decides, everyone follows
    outfile = os.path.join(ssl_mod.results_dir, tag + model.get_accuracy_filename())
    go = [bool(not save_results or overwrite or not os.path.exists(outfile))] if rank == 0 else [None]
    dist.broadcast_object_list(go, src=0, group=group)
    if not go[0]:
        if rank == 0:
            print('Aborting: SSL trial (' + model.get_accuracy_filename() + ') already completed , and overwrite is False.')
        return None
    bounds = block_bounds(len(trainsets), world)
    mine = list(model._trial_rows(trainsets[bounds[rank]:bounds[rank + 1]], labels))
    parts = [None] * world
    dist.all_gather_object(parts, mine, group=group)
    rows = [r for part in parts for r in part]
    if rank == 0:
        with_priors = model.class_priors is not None
        header = 'Number of labels,Accuracy,Accuracy with class priors,Class priors error' if with_priors else 'Number of labels,Accuracy'
        print('\nModel: ' + model.name + '\n\n' + header)
        for r in rows:
            print(r)
        if save_results:
            os.makedirs(ssl_mod.results_dir, exist_ok=True)
            with open(outfile, 'w') as f:
                f.write(header + '\n' + ''.join(r + '\n' for r in rows))
            print('Results File: ' + outfile)
    return rows
