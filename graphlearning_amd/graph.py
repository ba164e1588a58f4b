"""Graph object: the part of reference graphlearning/graph.py on the hot path
(`graph.__init__` :25-67, `degree_vector` :108-122, `degree_matrix` :210-233,
`laplacian` :469-513).  The reference's `__ccode_init__` (graph.py:69-84, 0.4-0.6 s per
construction at 70k nodes, never used by this path) is not executed."""
import sys
import numpy as np
from scipy import sparse


class graph:
    def __init__(self, W, labels=None, features=None, label_names=None, node_names=None):
        self.weight_matrix = sparse.csr_matrix(W)
        if getattr(W, '_glx_sym', None) is not None:
            self.weight_matrix._glx_sym = W._glx_sym      # a CONTENT fingerprint: utils.known_symmetric re-hashes the arrays, so an edited copy is simply unknown
        if getattr(W, '_glx_order', None) is not None:
            self.weight_matrix._glx_order = W._glx_order    # (a vertex order stays valid whatever happens to the values)
        self.labels = labels
        self.features = features
        self.num_nodes = W.shape[0]
        self.label_names = label_names
        self.node_names = node_names

    def degree_vector(self):
        """d_i = sum_j w_ij (row sums; reference graph.py:108-122)."""
        return self.weight_matrix * np.ones(self.num_nodes)

    def degree_matrix(self, p=1):
        """Sparse diagonal matrix of d^p (reference graph.py:210-233)."""
        d = self.degree_vector()
        return sparse.spdiags(d ** p, 0, self.num_nodes, self.num_nodes).tocsr()

    def laplacian(self, normalization='combinatorial'):
        """D-W, I-D^-1 W or I-D^-1/2 W D^-1/2 (reference graph.py:469-513)."""
        I = sparse.identity(self.num_nodes)
        D = self.degree_matrix()
        if normalization == 'combinatorial':
            L = D - self.weight_matrix
        elif normalization == 'randomwalk':
            L = I - self.degree_matrix(p=-1) * self.weight_matrix
        elif normalization == 'normalized':
            Dh = self.degree_matrix(p=-0.5)
            L = I - Dh * self.weight_matrix * Dh
        else:
            sys.exit('Invalid option for graph Laplacian normalization.')
        return L.tocsr()

    def reweight(self, idx, method='poisson', normalization='combinatorial', tau=0, X=None, alpha=2, zeta=1e7, r=0.1):
        """Reweight the graph more heavily near the labelled nodes `idx` (reference
        graph.py:368-466).  'poisson' solves one Poisson problem with the GPU conjugate-gradient
        solver (1-D right-hand side: numpy's pairwise-summed reductions are reproduced);
        'wnll' is a diagonal scaling; 'properly' scales by the distance to the nearest labelled point (all pairs on
        the GPU, glx_nearest_dist: cKDTree's distances bit for bit).  (The 'poisson' system is the singular graph Laplacian: its
        conjugate-gradient iterates amplify rounding, so the solve always uses the reference-order
        reductions -- the tolerance mode of ssl.laplace / ssl.randomwalk does not apply here.)"""
        from . import utils
        n = self.num_nodes
        if method == 'poisson':
            f = np.zeros(n)
            f[idx] = 1
            if normalization == 'combinatorial':
                f -= np.mean(f)
                L = self.laplacian()
            elif normalization == 'normalized':
                d = self.degree_vector() ** (0.5)
                c = np.sum(d * f) / np.sum(d)
                f -= c
                L = self.laplacian(normalization=normalization)
            else:
                sys.exit('Unsupported normalization ' + normalization + ' for graph.reweight.')
            w = utils.conjgrad(L, f, tol=1e-5)
            w -= np.min(w)
            w += 1e-5
            D = sparse.spdiags(w, 0, n, n).tocsr()
            return D * self.weight_matrix * D
        elif method == 'wnll':
            m = len(idx)
            a = np.ones((n,))
            a[idx] = n / m
            D = sparse.spdiags(a, 0, n, n).tocsr()
            return D * self.weight_matrix + self.weight_matrix * D
        elif method == 'properly':     # reference graph.py:448-462
            if X is None:
                sys.exit('Must provide data features X for properly weighted graph Laplacian.')
            from . import _hip
            rzeta = r / (zeta - 1) ** (1 / alpha)
            # `D, J = cKDTree(X[idx, :]).query(X)`: the distance to the nearest labelled point, all pairs on the GPU (the reference's bits)
            D = _hip.nearest_dist(X, idx)
            D[D < rzeta] = rzeta
            gamma = 1 + (r / D) ** alpha
            D = sparse.spdiags(gamma, 0, n, n).tocsr()
            return D * self.weight_matrix + self.weight_matrix * D
        else:
            sys.exit('Invalid reweighting method ' + method + '.')

    def page_rank(self, alpha=0.85, v=None, tol=1e-10, device=None):
        """PageRank vector by the power iteration u <- alpha P u + (1-alpha) v, P = W^T D^-1, from
        u = 1/n until max|u_new - u_old| <= tol (reference graph.py:1371-1412).  The host builds
        alpha*P with the reference's own scipy expressions; every sweep and the stop test run on the
        GPU (glx_affine_iterate).  Bit-identical to the reference: a row's products are added in
        the order scipy's matvec of that matrix adds them."""
        from . import _hip
        n = self.num_nodes
        u = np.ones((n,)) / n
        if v is None:
            v = np.ones((n,)) / n
        D = self.degree_matrix(p=-1)
        P = self.weight_matrix.T @ D
        aP = alpha * P              # `alpha*P@u` in the reference is (alpha*P)@u
        # csc_matvec adds a row's products column after column (ascending), which is the row order
        # csc -> csr conversion produces; a csr operand is used by csr_matvec in stored order
        A = aP.tocsr()
        self.page_rank_iters = 0
        if not (tol + 1 > tol):     # `err = tol+1; while err > tol` never enters the loop
            return u
        dev = _hip.DeviceGraph(A, dtype=np.float64, device=device)
        try:
            u, it, _ = dev.affine_iterate(u, b=(1 - alpha) * v, tol=tol)
        finally:
            dev.close()
        self.page_rank_iters = it
        return u

    def __ccode_init__(self):
        """Stored entries as (vertex, neighbour, weight) arrays sorted by vertex, the form the
        reference hands to its C extension (reference graph.py:69-84, same expressions: the order
        inside a vertex's block is whatever np.argsort's default sort leaves)."""
        I, J, V = sparse.find(self.weight_matrix)
        ind = np.argsort(I)
        self.I, self.J, self.V = I[ind], J[ind], V[ind]
        self.I = np.ascontiguousarray(self.I, dtype=np.int32)
        self.J = np.ascontiguousarray(self.J, dtype=np.int32)
        self.V = np.ascontiguousarray(self.V, dtype=np.float64)

    def plaplace(self, bdy_set, bdy_val, p, tol=1e-1, max_num_it=1e6, prog=False, fast=True, device=None):
        """Game-theoretic p-Laplace equation with Dirichlet data (reference graph.py:1177-1278).
        `fast=False` -- the Jacobi iteration of upper / lower barriers, lp_iterate_main of the
        reference's C extension -- runs on the GPU (glx_lp_iterate) and returns (uu+ul)/2 like the
        reference.  The reference's default `fast=True` is a Gauss-Seidel sweep (lip_iterate_main,
        c_code/lp_iterate.cpp:127-180): every vertex reads values updated earlier in the same sweep,
        an inherently sequential recurrence that has no parallel form with the same iterates."""
        from . import _hip, utils
        if fast:
            raise NotImplementedError('graph.plaplace(fast=True) is a sequential Gauss-Seidel sweep in the reference; '
                                      'pass fast=False for the Jacobi iteration, which runs on the GPU')
        if getattr(self, 'I', None) is None:
            self.__ccode_init__()
        n = self.num_nodes
        bdy_set, bdy_val = utils._boundary_handling(bdy_set, bdy_val)
        uu = np.max(bdy_val) * np.ones((n,))
        ul = np.min(bdy_val) * np.ones((n,))
        uu[bdy_set] = bdy_val
        ul[bdy_set] = bdy_val
        uu = np.ascontiguousarray(uu, dtype=np.float64)
        ul = np.ascontiguousarray(ul, dtype=np.float64)
        bdy_set = np.ascontiguousarray(bdy_set, dtype=np.int32)
        bdy_val = np.ascontiguousarray(bdy_val, dtype=np.float64)
        self.plaplace_iters = _hip.lp_iterate(uu, ul, self.J, self.I, self.V, bdy_set, bdy_val, p, int(max_num_it), float(tol),
                                              device=device)
        return (uu + ul) / 2

    def subgraph(self, ind):
        W = self.weight_matrix
        return graph(W[ind, :][:, ind])

    def isconnected(self):
        from scipy.sparse import csgraph
        return csgraph.connected_components(self.weight_matrix)[0] == 1


# the reference exposes the class as `graphlearning.graph` (its __init__ rebinds the name of this
# module to the class); `gl.graph.graph(W)` -- the module-style spelling -- keeps working too
graph.graph = graph
