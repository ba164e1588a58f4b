"""graphlearning_amd -- MI355X (gfx950) native kNN-graph + Poisson/Laplace label propagation.

Drop-in for the hot path of jwcalder/GraphLearning (reference v1.7.5):

    import graphlearning_amd as gl
    W = gl.weightmatrix.knn(X, 10)
    model = gl.ssl.poisson(W, solver='gradient_descent')
    pred = model.fit_predict(train_ind, train_labels)

Same names, arguments and error behaviour as `graphlearning.weightmatrix.knn/knnsearch`,
`graphlearning.graph.graph`, `graphlearning.ssl.poisson/laplace/poisson_mbo`,
`graphlearning.utils.conjgrad` ...; the iteration loops run as hand-written HIP kernels
behind the C-ABI of include/glx.h (libglx.so, loaded with ctypes).  There is no CPU
fallback: without the built library and a GPU the solvers raise.
"""
from . import utils
from .graph import graph          # like the reference: gl.graph(W) is the class (gl.graph.graph also works)
from . import trainsets
from . import weightmatrix
from . import ssl
from ._hip import GlxError

__version__ = '0.1.0'
