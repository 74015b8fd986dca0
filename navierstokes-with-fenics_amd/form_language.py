"""The slice of dolfin's form language the reference's CALLERS use on the results of the hot path:
arithmetic on Functions / Constants / Expressions, ``dot`` / ``inner`` / ``grad`` / ``.T`` /
indexing, ``FacetNormal``, the measures ``ds`` / ``dx`` (``Measure("ds", domain, subdomain_data)``,
``ds(subdomain_id=...)``), ``assemble`` of a functional and ``project`` onto a P1 space.

Reference call sites: demo/dfg_benchmark.py:44-66 (drag / lift from the surface traction),
demo/gravity_driven_flow.py:38-67 and tests/test_stationary_solvers.py:84-110 (Bernoulli potential
projected on CG1, total mass flux over the boundary).

This is post-processing, not the per-step path: expressions are small trees evaluated on the
host with numpy at quadrature points -- Gauss points of the marked boundary facets (``ds``: their
adjacent cells supply the finite element values and gradients) or of all cells (``dx``).  On affine
cells every integrand the callers build is a polynomial of degree <= 4, the rules used are exact for
degree 6 (dx) / 7 (ds in 2D) / 5 (ds in 3D).  The mass solve of ``project`` runs on the GPU
(nsfem_mass_solve), like the projections of the initial conditions.  The device kernel behind
``nsfem_boundary_force`` computes the same traction / flux functionals without this layer
(bench.py, ProblemBase._compute_boundary_force)."""
import numpy as np

import fem_host


# ------------------------------------------------------------------------------- tree nodes
class Operand:
    """arithmetic mix-in: every object that may appear in an expression derives from it"""
    __array_priority__ = 1000          # numpy scalars / arrays defer to our operators

    def __add__(self, o):
        return _Binary("+", self, o)

    def __radd__(self, o):
        return _Binary("+", o, self)

    def __sub__(self, o):
        return _Binary("-", self, o)

    def __rsub__(self, o):
        return _Binary("-", o, self)

    def __mul__(self, o):
        if isinstance(o, Measure):
            return Form(self, o)
        return _Binary("*", self, o)

    def __rmul__(self, o):
        return _Binary("*", o, self)

    def __truediv__(self, o):
        return _Binary("/", self, o)

    def __rtruediv__(self, o):
        return _Binary("/", o, self)

    def __pow__(self, o):
        return _Binary("**", self, o)

    def __neg__(self):
        return _Binary("*", -1.0, self)

    def __pos__(self):
        return self

    def __getitem__(self, index):
        return _Index(self, index)

    @property
    def T(self):
        return _Transpose(self)


class _Binary(Operand):
    def __init__(self, op, a, b):
        self.op, self.a, self.b = op, a, b


class _Index(Operand):
    def __init__(self, a, index):
        self.a, self.index = a, index if isinstance(index, tuple) else (index, )


class _Transpose(Operand):
    def __init__(self, a):
        self.a = a


class _Contract(Operand):
    """dot (last axis of a with first value axis of b) or inner (all value axes)"""

    def __init__(self, a, b, full):
        self.a, self.b, self.full = a, b, full


class _Grad(Operand):
    def __init__(self, f):
        self.f = f


class _Call(Operand):
    def __init__(self, fn, a):
        self.fn, self.a = fn, a


class FacetNormal(Operand):
    """outward unit normal of the boundary facets (``dolfin.FacetNormal(mesh)``)"""

    def __init__(self, mesh):
        self.mesh = mesh


def dot(a, b):
    return _Contract(a, b, False)


def inner(a, b):
    return _Contract(a, b, True)


def grad(f):
    return _Grad(f)


def sqrt(a):
    return _Call(np.sqrt, a)


# ------------------------------------------------------------------------------- measures, forms
class Measure:
    def __init__(self, kind, domain=None, subdomain_data=None, subdomain_id=None):
        assert kind in ("ds", "dx"), "only boundary (ds) and cell (dx) measures are provided"
        self.kind, self.domain = kind, domain
        self.subdomain_data, self.subdomain_id = subdomain_data, subdomain_id

    def __call__(self, subdomain_id=None, domain=None, subdomain_data=None):
        return Measure(self.kind, domain if domain is not None else self.domain,
                       subdomain_data if subdomain_data is not None else self.subdomain_data,
                       subdomain_id if subdomain_id is not None else self.subdomain_id)

    def __rmul__(self, integrand):
        return Form(integrand, self)


ds = Measure("ds")
dx = Measure("dx")


class Form:
    """sum of (integrand, measure) pairs; ``assemble`` turns it into a number"""

    def __init__(self, integrand, measure):
        self.terms = [(integrand, measure)]

    def __add__(self, other):
        out = Form.__new__(Form)
        out.terms = self.terms + other.terms
        return out

    def __neg__(self):
        out = Form.__new__(Form)
        out.terms = [(_Binary("*", -1.0, i), m) for i, m in self.terms]
        return out

    def __sub__(self, other):
        return self + (-other)


# ------------------------------------------------------------------------------- evaluation
class _Points:
    """evaluation points of one integral: physical coordinates, adjacent cells, reference
    coordinates inside them, outward normals (ds only), weights incl. the measure"""

    def __init__(self, mesh, cells, ref, X, weights, normals=None):
        self.mesh, self.cells, self.ref, self.X, self.weights, self.normals = mesh, cells, ref, X, weights, normals
        self._cache = {}

    def cell_jacobian_inverse_T(self):
        if "jit" not in self._cache:
            x = self.mesh.coords[self.mesh.cells[self.cells].astype(np.int64)]
            J = np.transpose(x[:, 1:] - x[:, :1], (0, 2, 1))          # columns = edge vectors
            self._cache["jit"] = np.transpose(np.linalg.inv(J), (0, 2, 1))
        return self._cache["jit"]


def _p2_shape_grad(ref):
    """reference gradients of the P2 basis at points ref [N, dim] -> [N, nloc, dim]"""
    dim = ref.shape[1]
    lam = fem_host._p1_shape(ref)
    dlam = np.zeros((dim + 1, dim))
    dlam[0, :] = -1.0
    dlam[1:, :] = np.eye(dim)
    out = [(4.0 * lam[:, i] - 1.0)[:, None] * dlam[i][None, :] for i in range(dim + 1)]
    out += [4.0 * (lam[:, a, None] * dlam[b][None, :] + lam[:, b, None] * dlam[a][None, :])
            for a, b in fem_host._EDGE_PAIRS[dim]]
    return np.stack(out, axis=1)


def _fe_data(f):
    """(dof map, field kind, coefficient vector) of a finite element function object"""
    if hasattr(f, "_solver") and hasattr(f, "field"):                 # fem_function.DeviceFunction
        return f._solver._dofmap, f.field, f.vector()
    space = f.function_space()                                        # fem_spaces.Function
    assert space.kind in ("velocity", "pressure"), "split() a mixed function before using it in a form"
    return space.dofmap, space.kind, f.vector()


def _is_fe_function(f):
    return (hasattr(f, "_solver") and hasattr(f, "field")) or hasattr(f, "function_space")


def _fe_value(f, pts, gradient=False):
    dm, kind, values = _fe_data(f)
    assert dm.mesh is pts.mesh or dm.mesh.num_cells() == pts.mesh.num_cells(), "function lives on another mesh"
    dim = dm.dim
    if kind == "velocity":
        nodal = values.reshape(-1, dim)[dm.p2_dofmap[pts.cells].astype(np.int64)]     # [N, nloc, dim]
        if not gradient:
            return np.einsum("nk,nka->na", fem_host._p2_shape(pts.ref), nodal)
        g = np.einsum("nab,nkb->nka", pts.cell_jacobian_inverse_T(), _p2_shape_grad(pts.ref))
        return np.einsum("nkb,nka->nab", g, nodal)                                    # d_b u_a
    nodal = values[dm.p1_dofmap[pts.cells].astype(np.int64)]                          # [N, dim + 1]
    if not gradient:
        return np.einsum("nk,nk->n", fem_host._p1_shape(pts.ref), nodal)
    dlam = np.zeros((dim + 1, dim))
    dlam[0, :] = -1.0
    dlam[1:, :] = np.eye(dim)
    g = np.einsum("nab,kb->nka", pts.cell_jacobian_inverse_T(), dlam)
    return np.einsum("nka,nk->na", g, nodal)


def _align(a, b):
    """broadcast a scalar-valued array against a tensor-valued one"""
    while a.ndim < b.ndim:
        a = a[..., None]
    while b.ndim < a.ndim:
        b = b[..., None]
    return a, b


def _evaluate(node, pts):
    """value of ``node`` at the points: array [N] (scalar), [N, d] (vector) or [N, d, d] (tensor)"""
    n = pts.X.shape[0]
    if isinstance(node, (int, float, np.floating, np.integer)):
        return np.full(n, float(node))
    if isinstance(node, _Binary):
        a, b = _align(_evaluate(node.a, pts), _evaluate(node.b, pts))
        if node.op == "+":
            return a + b
        if node.op == "-":
            return a - b
        if node.op == "*":
            return a * b
        if node.op == "/":
            return a / b
        return a ** b
    if isinstance(node, _Index):
        return _evaluate(node.a, pts)[(slice(None), ) + node.index]
    if isinstance(node, _Transpose):
        a = _evaluate(node.a, pts)
        assert a.ndim == 3, ".T needs a tensor-valued expression"
        return np.transpose(a, (0, 2, 1))
    if isinstance(node, _Contract):
        a, b = _evaluate(node.a, pts), _evaluate(node.b, pts)
        if node.full:
            assert a.shape == b.shape
            return (a * b).reshape(n, -1).sum(axis=1) if a.ndim > 1 else a * b
        assert a.ndim >= 2 and b.ndim >= 2, "dot needs vector / tensor arguments"
        return np.einsum("n...i,ni...->n...", a, b)
    if isinstance(node, _Grad):
        return _gradient(node.f, pts)
    if isinstance(node, _Call):
        return node.fn(_evaluate(node.a, pts))
    if isinstance(node, FacetNormal):
        assert pts.normals is not None, "FacetNormal outside of a boundary integral"
        return pts.normals
    if _is_fe_function(node):
        return _fe_value(node, pts)
    if hasattr(node, "eval_at"):                                      # Constant / Expression / UserExpression
        return np.asarray(node.eval_at(pts.X), dtype=np.float64)
    raise TypeError("cannot evaluate %r inside a form" % (node, ))


def _gradient(f, pts):
    if _is_fe_function(f):
        return _fe_value(f, pts, gradient=True)
    if isinstance(f, _Binary) and f.op in "+-":
        a, b = _gradient(f.a, pts), _gradient(f.b, pts)
        return a + b if f.op == "+" else a - b
    if isinstance(f, _Binary) and f.op == "*" and isinstance(f.a, (int, float)):
        return f.a * _gradient(f.b, pts)
    if isinstance(f, _Binary) and f.op == "*" and isinstance(f.b, (int, float)):
        return f.b * _gradient(f.a, pts)
    raise TypeError("grad() is provided for finite element functions and their linear combinations")


def _facet_points(mesh, facet_ids):
    """Gauss points on the given boundary facets, pulled back to their adjacent cells"""
    dim = mesh._dim
    facet_ids = np.asarray(facet_ids, dtype=np.int64)
    cells = mesh.facet_cell[facet_ids].astype(np.int64)
    xf = mesh.coords[mesh.facets[facet_ids].astype(np.int64)]          # [nf, dim, dim]
    if dim == 2:
        g, w = np.polynomial.legendre.leggauss(4)
        lam = np.stack([0.5 * (1.0 - g), 0.5 * (1.0 + g)], axis=1)
        wq = 0.5 * w
        measure = np.linalg.norm(xf[:, 1] - xf[:, 0], axis=1)
    else:
        p, wt = fem_host.conical_rule(3, 2)
        lam = np.concatenate([(1.0 - p.sum(axis=1))[:, None], p], axis=1)
        wq = 2.0 * wt
        measure = 0.5 * np.linalg.norm(np.cross(xf[:, 1] - xf[:, 0], xf[:, 2] - xf[:, 0]), axis=1)
    X = np.einsum("qv,fvd->fqd", lam, xf)                              # [nf, q, dim]
    xc = mesh.coords[mesh.cells[cells].astype(np.int64)]
    J = np.transpose(xc[:, 1:] - xc[:, :1], (0, 2, 1))
    ref = np.einsum("fab,fqb->fqa", np.linalg.inv(J), X - xc[:, None, 0, :])
    nq = lam.shape[0]
    normals = np.repeat(mesh.facet_normals(facet_ids), nq, axis=0)
    weights = (measure[:, None] * wq[None, :]).ravel()
    return _Points(mesh, np.repeat(cells, nq), ref.reshape(-1, dim), X.reshape(-1, dim), weights, normals)


def _cell_points(mesh, c0, c1):
    dim = mesh._dim
    p, wt = fem_host.conical_rule(4, dim)
    cells = np.arange(c0, c1, dtype=np.int64)
    x = mesh.coords[mesh.cells[cells].astype(np.int64)]
    lam = fem_host._p1_shape(p)
    X = np.einsum("qv,cvd->cqd", lam, x)
    det = np.abs(np.linalg.det(x[:, 1:] - x[:, :1]))
    nq = p.shape[0]
    return _Points(mesh, np.repeat(cells, nq), np.tile(p, (cells.size, 1)), X.reshape(-1, dim),
                   (det[:, None] * wt[None, :]).ravel())


def _mesh_of(node):
    """the mesh an expression lives on (from its finite element functions / facet normal)"""
    if isinstance(node, FacetNormal):
        return node.mesh
    if _is_fe_function(node):
        return _fe_data(node)[0].mesh
    for attr in ("a", "b", "f"):
        child = getattr(node, attr, None)
        if child is not None and not isinstance(child, (int, float, str, tuple)):
            m = _mesh_of(child)
            if m is not None:
                return m
    return None


def assemble(form):
    """value of a functional  sum_k int integrand_k d(measure_k)"""
    assert isinstance(form, Form), "assemble() takes integrand * measure"
    total = 0.0
    for integrand, measure in form.terms:
        mesh = measure.domain if measure.domain is not None else _mesh_of(integrand)
        assert mesh is not None, "the measure needs a domain"
        if measure.kind == "ds":
            if measure.subdomain_id is None:
                facets = np.nonzero(mesh.facet_on_boundary)[0]
            else:
                assert measure.subdomain_data is not None, "ds(subdomain_id) needs subdomain_data"
                facets = measure.subdomain_data.facets_with_id(measure.subdomain_id)
                facets = facets[mesh.facet_on_boundary[facets]]
            if facets.size == 0:
                continue
            pts = _facet_points(mesh, facets)
            v = _evaluate(integrand, pts)
            assert v.ndim == 1, "assemble() needs a scalar integrand"
            total += float(v @ pts.weights)
        else:
            assert measure.subdomain_id is None, "cell subdomains are not provided"
            nq = fem_host.conical_rule(4, mesh._dim)[0].shape[0]
            chunk = max(1, 2_000_000 // nq)
            for c0 in range(0, mesh.num_cells(), chunk):
                pts = _cell_points(mesh, c0, min(mesh.num_cells(), c0 + chunk))
                v = _evaluate(integrand, pts)
                assert v.ndim == 1, "assemble() needs a scalar integrand"
                total += float(v @ pts.weights)
    return total


# ------------------------------------------------------------------------------- projection
class LagrangeSpaceRequest:
    """``dolfin.FunctionSpace(mesh, "CG", degree)``: resolved against a dof map when used"""

    def __init__(self, mesh, family, degree):
        assert family in ("CG", "Lagrange", "P") and degree in (1, 2), "P1 / P2 Lagrange spaces only"
        self._mesh, self.degree = mesh, degree

    def mesh(self):
        return self._mesh


def _solver_of(node):
    if hasattr(node, "_solver"):
        return node._solver
    for attr in ("a", "b", "f"):
        child = getattr(node, attr, None)
        if child is not None and not isinstance(child, (int, float, str, tuple)):
            s = _solver_of(child)
            if s is not None:
                return s
    return None


def project_expression(expression, space):
    """L2 projection of a scalar form expression onto the P1 space of the mesh:
    b_i = int expression phi_i dx (host quadrature), M x = b on the device (nsfem_mass_solve).
    Returns a node-centred field object (``rename`` / file output)."""
    import _native as nat
    from fem_function import HostField
    assert isinstance(space, LagrangeSpaceRequest) and space.degree == 1, "projection target: FunctionSpace(mesh, 'CG', 1)"
    solver = _solver_of(expression)
    assert solver is not None, "the expression must contain a solver field (velocity / pressure)"
    dm = solver._dofmap
    mesh = dm.mesh
    dim = mesh._dim
    b = np.zeros(dm.n_p1)
    p, _ = fem_host.conical_rule(4, dim)
    phi = fem_host._p1_shape(p)                                       # [q, dim + 1]
    nq = p.shape[0]
    chunk = max(1, 2_000_000 // nq)
    for c0 in range(0, mesh.num_cells(), chunk):
        c1 = min(mesh.num_cells(), c0 + chunk)
        pts = _cell_points(mesh, c0, c1)
        v = _evaluate(expression, pts)
        assert v.ndim == 1, "project() onto CG1 needs a scalar expression"
        be = np.einsum("cq,qk->ck", (v * pts.weights).reshape(c1 - c0, nq), phi)
        np.add.at(b, dm.p1_dofmap[c0:c1].astype(np.int64).ravel(), be.ravel())
    x = solver._ctx.mass_solve(nat.PRESSURE, b)
    return HostField(mesh, "projection", "Node", x[dm.p1_vertex_node])
