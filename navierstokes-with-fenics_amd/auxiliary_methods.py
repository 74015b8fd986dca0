"""Mesh helper functions under the reference's module name (source/auxiliary_methods.py:8-67):
``boundary_normal(mesh, facet_markers, bndry_id)`` -- the common outward unit normal of a flat
boundary part (asserts flatness, as the reference does) -- and
``extract_all_boundary_markers(mesh, mesh_function)`` -- the set of marker ids found on the
boundary.  Implemented in ``fem_host`` on the dolfin-free mesh containers."""
from fem_host import boundary_normal, extract_all_boundary_markers  # noqa: F401

__all__ = ["boundary_normal", "extract_all_boundary_markers"]
