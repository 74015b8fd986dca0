"""Why is the dictionary of the gradient block G = (grad p, w) not bit-exact on a binary lattice while D^T is?
Export both, compare rows of equal (parity class, interior) position."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "navierstokes-with-fenics_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _native as nat
from gpu_common import box, context
n = 64
mesh, dm, _ = box(n, n)
ctx = context(mesh, dm)
for name, op in (("GRAD", nat.OP_GRAD), ("DIVT", nat.OP_DIVT)):
    A = ctx.operator_csr(op).tocsr()
    W = 2 * n + 1
    nrow = A.shape[0] // 2 if A.shape[0] == 2 * dm.n_p2 else A.shape[0]
    print(name, A.shape, A.nnz)
    # rows of P2 node (i, j), component 0 -> row 2 * node?  (block rows 2 x 1: exported scalar rows 2 node + c)
    def row(node, c):
        r = 2 * node + c
        return A.indices[A.indptr[r]:A.indptr[r + 1]], A.data[A.indptr[r]:A.indptr[r + 1]]
    for (pi, pj) in ((0, 0), (1, 0), (0, 1), (1, 1)):
        ref = None
        worst = 0.0
        nz_tiny = 0
        for j in range(8 + pj, W - 8, 2):
            for i in range(8 + pi, W - 8, 2):
                node = j * W + i
                for c in (0, 1):
                    cols, vals = row(node, c)
                    key = (c,)
                    if ref is None: ref = {}
                    if key not in ref:
                        ref[key] = (cols - cols[0], vals.copy()); continue
                    rc, rv = ref[key]
                    assert np.array_equal(cols - cols[0], rc)
                    d = np.abs(vals - rv).max()
                    worst = max(worst, d)
                    nz_tiny += int(((np.abs(vals) > 0) & (np.abs(vals) < 1e-12 * np.abs(rv).max())).sum())
        print("  class", (pi, pj), "max |row - first row|", worst, "tiny nonzeros", nz_tiny, "ref vals", ref[(0,)][1][:8])
ctx.close()
