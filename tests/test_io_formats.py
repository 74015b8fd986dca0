"""I/O formats either side of the hot path: gmsh .msh ingest and XDMF
field output.  CPU tests: format round trips and a hand-written MSH 4.1 fixture; the GPU test
drives a problem with output switched on and reads the file back."""
import os

import numpy as np
import pytest

from fem_function import HostField
from fem_mesh import FacetMarkers, TaylorHoodDofMap, rectangle_mesh
from mesh_io import read_msh, write_msh
from xdmf_io import XDMFFile, read_xdmf

HERE = os.path.dirname(os.path.abspath(__file__))


def test_msh_v41_fixture():
    mesh, marks, names, cell_phys = read_msh(os.path.join(HERE, "golden", "square_v41.msh"))
    assert mesh.num_vertices() == 5 and mesh.num_cells() == 4
    assert names == {"inlet": (1, 1), "wall": (1, 2), "fluid": (2, 3)}
    assert (cell_phys == 3).all()
    mid = mesh.edge_midpoints()
    left = np.nonzero(marks.values == 1)[0]
    assert left.size == 1 and np.allclose(mid[left[0]], [0.0, 0.5])
    walls = mid[marks.values == 2]
    assert walls.shape[0] == 3 and np.allclose(sorted(walls[:, 0]), [0.5, 0.5, 1.0])
    x = mesh.coords[mesh.cells]
    d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
    det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
    assert (det > 0).all() and abs(0.5 * det.sum() - 1.0) < 1e-14


def test_msh_v22_round_trip_of_the_dfg_mesh(tmp_path):
    from grid_generator import DFGBoundaryMarkers, dfg_channel
    mesh, marks = dfg_channel(2, 1)
    names = {m.name: (1, m.value) for m in DFGBoundaryMarkers}
    names["fluid"] = (2, 1)
    path = str(tmp_path / "dfg.msh")
    write_msh(path, mesh, marks, names)
    mesh2, marks2, names2, _ = read_msh(path)
    assert names2 == names
    assert np.array_equal(mesh2.coords, mesh.coords)
    assert np.array_equal(np.sort(mesh2.cells, axis=1), np.sort(mesh.cells, axis=1))
    assert np.array_equal(mesh2.edges, mesh.edges) and np.array_equal(marks2.values, marks.values)
    dm = TaylorHoodDofMap(mesh2)                      # the ingested mesh feeds the solver's dof map
    assert dm.n_p1 == mesh.num_vertices() and dm.n_p2 == mesh.num_vertices() + mesh.num_edges()


def test_msh_reader_rejects_what_it_cannot_represent(tmp_path):
    p = tmp_path / "quad.msh"
    p.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n4\n1 0 0 0\n2 1 0 0\n3 1 1 0\n4 0 1 0\n"
                 "$EndNodes\n$Elements\n1\n1 3 2 1 1 1 2 3 4\n$EndElements\n")
    with pytest.raises(ValueError):
        read_msh(str(p))


@pytest.mark.parametrize("encoding", ["binary", "xml"])
def test_xdmf_writer_round_trip(tmp_path, encoding):
    mesh = rectangle_mesh((0.0, 0.0), (2.0, 1.0), 4, 3)
    f = XDMFFile(str(tmp_path / "out.xdmf"), encoding=encoding)
    rng = np.random.default_rng(0)
    written = []
    for step, t in enumerate((0.0, 0.5, 1.25)):
        u = rng.standard_normal((mesh.num_vertices(), 2))
        p = rng.standard_normal(mesh.num_vertices())
        w = rng.standard_normal(mesh.num_cells())
        f.write(HostField(mesh, "velocity", "Node", u), t)
        f.write(HostField(mesh, "pressure", "Node", p), t)
        f.write(HostField(mesh, "vorticity", "Cell", w), t)
        written.append((u, p, w))
        back = read_xdmf(str(tmp_path / "out.xdmf"))      # flush_output: valid after every write
        assert len(back["times"]) == step + 1
    back = read_xdmf(str(tmp_path / "out.xdmf"))
    assert back["times"] == [0.0, 0.5, 1.25]
    assert np.array_equal(back["cells"], mesh.cells) and np.array_equal(back["coords"], mesh.coords)
    assert back["centers"] == {"velocity": "Node", "pressure": "Node", "vorticity": "Cell"}
    for i, (u, p, w) in enumerate(written):
        assert np.array_equal(back["fields"]["velocity"][i][:, :2], u)
        assert np.array_equal(back["fields"]["pressure"][i], p)
        assert np.array_equal(back["fields"]["vorticity"][i], w)


@pytest.mark.gpu
def test_problem_writes_xdmf_with_vorticity_and_pressure_gradient(tmp_path):
    """Poiseuille start-up channel written every 2 steps: vertex values in the file equal the
    solver's fields; vorticity / pressure gradient of the exact parabolic profile are exact."""
    import dlfn_compat as dlfn
    from auxiliary_classes import EquationCoefficientHandler
    from grid_generator import HyperRectangleBoundaryMarkers as M, hyper_rectangle
    from ns_ipcs_solver import IPCSSolver
    from ns_problem import InstationaryProblem, PressureBCType, VelocityBCType

    class Channel(InstationaryProblem):
        def __init__(self, main_dir):
            super().__init__(main_dir, start_time=0.0, end_time=1.0, desired_start_time_step=0.05,
                             n_max_steps=4)
            self._problem_name = "Channel"
            self._output_frequency = 2
            self._postprocessing_frequency = 2
            self.set_solver_class(IPCSSolver)

        def setup_mesh(self):
            self._mesh, self._boundary_markers = hyper_rectangle((0.0, 0.0), (2.0, 1.0), (8, 4))

        def set_equation_coefficients(self):
            self._coefficient_handler = EquationCoefficientHandler(Re=1.0)

        def set_initial_conditions(self):
            self._initial_conditions = {"velocity": dlfn.Expression(("6.0*x[1]*(1.0-x[1])", "0.0"), degree=2),
                                        "pressure": dlfn.Expression("12.0*(2.0-x[0])", degree=1)}

        def set_boundary_conditions(self):
            inlet = dlfn.Expression(("6.0*x[1]*(1.0-x[1])", "0.0"), degree=2)
            self._bcs = ((VelocityBCType.function, M.left.value, inlet),
                         (VelocityBCType.no_slip, M.bottom.value, None),
                         (VelocityBCType.no_slip, M.top.value, None),
                         (PressureBCType.constant, M.right.value, 0.0))

        def postprocess_solution(self):
            self._add_to_field_output(self._compute_pressure_gradient())
            self._add_to_field_output(self._compute_vorticity())

    problem = Channel(str(tmp_path))
    problem.solve_problem()
    files = [f for f in os.listdir(tmp_path / "results") if f.endswith(".xdmf")]
    assert files == ["Channel_Re1.000e+00.xdmf"]
    back = read_xdmf(str(tmp_path / "results" / files[0]))
    assert back["times"] == pytest.approx([0.0, 0.1, 0.2])
    solver = problem._get_solver()
    mesh = solver._mesh
    u = solver.solution.split()[0]
    vertex_u = np.array([u(x) for x in mesh.coords])
    assert np.abs(back["fields"]["velocity"][-1][:, :2] - vertex_u).max() < 1e-12
    X = mesh.coords
    assert np.abs(back["fields"]["velocity"][-1][:, 0] - 6 * X[:, 1] * (1 - X[:, 1])).max() < 1e-8
    assert np.abs(back["fields"]["pressure"][-1] - 12.0 * (2.0 - X[:, 0])).max() < 1e-6
    gp = back["fields"]["pressure gradient"][-1]
    assert back["centers"]["pressure gradient"] == "Cell" and np.abs(gp[:, 0] + 12.0).max() < 1e-6
    xc = X[mesh.cells].mean(axis=1)
    w = back["fields"]["vorticity"][-1]                  # -du_x/dy = -6 (1 - 2 y), linear: mean = centroid value
    assert np.abs(w + 6.0 * (1.0 - 2.0 * xc[:, 1])).max() < 1e-7


def test_external_mesh_entry_points(tmp_path):
    """grid_generator.channel_with_cylinder & co. (reference source/grid_generator.py:406-455):
    a supplied .msh is picked up from below the working directory; without it the DFG channel
    falls back to the in-repo triangulation and the other two say what is missing."""
    import grid_generator as gg
    mesh, marks, names = gg.channel_with_cylinder(2, 0)
    assert names["cylinder"] == gg.DFGBoundaryMarkers.cylinder.value and mesh.num_cells() > 100
    mesh_b, marks_b, names_b = gg.blasius_plate(8)                  # in-repo stand-in: internal plate line
    assert set(names_b) == {"inlet", "outlet", "bottom", "top", "plate"}
    plate = marks_b.facets_with_id(names_b["plate"])
    assert plate.size == 8 and not mesh_b.facet_on_boundary[plate].any()
    os.makedirs("meshes")
    write_msh(os.path.join("meshes", "DFGBenchmark.msh"), mesh, marks,
              {"inlet": (1, 1), "cylinder": (1, 5), "fluid": (2, 1)})
    mesh2, marks2, names2 = gg.channel_with_cylinder()
    assert names2 == {"inlet": 1, "cylinder": 5}
    assert mesh2.num_cells() == mesh.num_cells() and np.array_equal(marks2.values, marks.values)


def test_write_boundary_markers(tmp_path):
    """ProblemBase.write_boundary_markers (reference source/ns_problem.py:329-348)"""
    import xml.etree.ElementTree as ET
    from grid_generator import open_hyper_cube
    from ns_problem import ProblemBase
    prob = ProblemBase(str(tmp_path))
    prob._problem_name = "OpenCube"
    prob._mesh, prob._boundary_markers = open_hyper_cube(2, 8, (("top", (0.5, 1.0), 0.5), ))
    path = prob.write_boundary_markers()
    root = ET.parse(path).getroot()
    grid = root.find("Domain").find("Grid")
    n = int(grid.find("Topology").get("NumberOfElements"))
    assert n == 4 * 8
    vals = np.array(grid.find("Attribute").find("DataItem").text.split(), dtype=int)
    assert vals.size == n and set(vals.tolist()) == {1, 2, 3, 4, 7}


@pytest.mark.gpu
def test_velocity_potential_post_processing():
    """ProblemBase._compute_stream_potential (reference source/ns_problem.py:105-176): for the
    irrotational field u = grad(x^2 - y^2)/2 = (x, -y) in a box with a no-slip-marked bottom the
    weak problem (grad phi, grad psi) = (div u, psi) - (n . u, psi)_Gamma_other, phi = 0 on the
    bottom, has the solution of  -lap phi = 0,  d phi / dn = n . u: phi = (x^2 - y^2)/2 (zero on
    y = 0 only up to the x^2/2 part, so the discrete answer is compared with a host solve of the
    same weak form)."""
    import _native as nat
    import fem_oracle as fo
    from grid_generator import HyperCubeBoundaryMarkers as M, hyper_cube
    from ns_problem import ProblemBase, VelocityBCType

    class Holder:
        pass

    mesh, marks = hyper_cube(2, 8)
    dm = TaylorHoodDofMap(mesh)
    ctx = nat.NsfemContext(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap, dm.n_p2, dm.n_p1)
    X = dm.p2_coords
    u = np.stack([X[:, 0], -X[:, 1]], axis=1).ravel()
    ctx.set_state(nat.U0, u)
    solver = Holder()
    solver._dofmap, solver._ctx = dm, ctx
    prob = ProblemBase(None)
    prob._mesh, prob._boundary_markers = mesh, marks
    prob._bcs = ((VelocityBCType.no_slip, M.bottom.value, None), (VelocityBCType.no_normal_flux, M.left.value, None))
    prob._get_solver = lambda: solver
    from fem_function import DeviceFunction
    solver.solution = None
    prob._get_velocity = lambda: DeviceFunction(solver, "velocity", nat.U0)
    phi = prob._compute_stream_potential()
    # host solve of the same weak form with the oracle's operators
    s = fo.Space(mesh.coords, mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
    rhs = s.divergence() @ u
    for bid, (axis, value, sign) in ((M.right.value, (0, 1.0, 1.0)), (M.top.value, (1, 1.0, 1.0))):
        facets = marks.facets_with_id(bid)
        e = mesh.facets[facets]
        for a, b in e:                                  # n . u = +x on the right (= 1), -y on top (= -1)
            flux = 1.0 if bid == M.right.value else -1.0
            length = np.linalg.norm(mesh.coords[a] - mesh.coords[b])
            rhs[a] -= 0.5 * flux * length
            rhs[b] -= 0.5 * flux * length
    bottom = np.unique(dm.facet_p1_nodes(marks.facets_with_id(M.bottom.value)))
    A = fo.apply_dirichlet_rows(s.stiffness_p1(), bottom).tolil()
    rhs[bottom] = 0.0
    ref = fo.spla.spsolve(A.tocsc(), rhs)
    assert np.abs(phi.values - ref).max() < 1e-9 * np.abs(ref).max()
    assert prob._get_boundary_conditions_map() == {VelocityBCType.no_slip: (M.bottom.value, ),
                                                   VelocityBCType.no_normal_flux: (M.left.value, )}
    ctx.close()


def test_grid_tools_xdmf_mesh_conversion_and_ingest(tmp_path):
    """grid_tools.generate_xdmf_mesh (reference source/grid_tools.py:70-121): .msh -> the XDMF
    pair (cells + cell_markers, boundary facets + facet_markers, XML light data), read back by
    read_xdmf_mesh; grid_generator picks the pair up when the .msh itself is not supplied and
    takes the marker names from the .geo file."""
    import grid_generator as gg
    from grid_tools import generate_xdmf_mesh, read_xdmf_mesh
    mesh, marks = gg.dfg_channel(2, 0)
    os.makedirs("meshes")
    geo = os.path.join("meshes", "DFGBenchmark.geo")
    with open(geo, "w") as fh:
        fh.write('Physical Curve("inlet", 1) = {1};\nPhysical Curve("outlet", 2) = {2};\n'
                 'Physical Curve("bottom", 3) = {3};\nPhysical Curve("top", 4) = {4};\n'
                 'Physical Curve("cylinder", 5) = {5, 6};\nPhysical Surface("fluid", 1) = {1};\n')
    with pytest.raises(RuntimeError):                      # no gmsh here and no .msh yet
        generate_xdmf_mesh(geo)
    msh = os.path.join("meshes", "DFGBenchmark.msh")
    write_msh(msh, mesh, marks, {"inlet": (1, 1), "cylinder": (1, 5), "fluid": (2, 1)})
    xdmf_file, facet_file = generate_xdmf_mesh(geo)
    assert xdmf_file.endswith("DFGBenchmark.xdmf") and facet_file.endswith("DFGBenchmark_facet_markers.xdmf")
    assert 'Name="cell_markers"' in open(xdmf_file).read()
    assert 'Name="facet_markers"' in open(facet_file).read() and "Polyline" in open(facet_file).read()
    mesh2, marks2, cell_markers = read_xdmf_mesh(xdmf_file, facet_file)
    assert np.array_equal(np.sort(mesh2.cells, axis=1), np.sort(mesh.cells, axis=1))      # (write_msh orients)
    assert np.abs(mesh2.coords - mesh.coords).max() == 0.0
    assert np.array_equal(marks2.values, marks.values) and (cell_markers == 1).all()
    os.remove(msh)                                          # only the converted files are left
    mesh3, marks3, names = gg.channel_with_cylinder()
    assert names == {"inlet": 1, "outlet": 2, "bottom": 3, "top": 4, "cylinder": 5}
    assert np.array_equal(marks3.values, marks.values) and mesh3.num_cells() == mesh.num_cells()
