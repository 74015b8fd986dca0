"""N > 1 path on CPU: two gloo ranks (torch.distributed, world_size 2) run a strip-partitioned
SpMV and a Jacobi-preconditioned CG with the halo ranges and ownership masks of
``partition.StripPartition`` -- exactly the exchange pattern the device library performs over
RCCL (contiguous send-up / send-down ranges, all-reduced partial dot products) -- and are
checked against the undistributed oracle operators.  Local operators come from the oracle
assembled on each rank's local mesh (own rows + one ghost row): owned rows must equal the
global rows without any assembly communication."""
import os
import socket

import numpy as np
import pytest

import fem_oracle as fo
from fem_mesh import TaylorHoodDofMap, rectangle_mesh
from partition import GHOST, StripPartition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _exchange(dist, torch, part_halo, vec, width, rank, size):
    """fill the ghost ranges of ``vec`` (entries per node = width) from the neighbours"""
    reqs, bufs = [], []
    t = torch.from_numpy(vec)

    def rng(key):
        off, cnt = part_halo[key]
        return slice(off * width, (off + cnt) * width)
    if rank + 1 < size:
        reqs.append(dist.isend(t[rng("send_up")].clone(), rank + 1))
        b = torch.empty(part_halo["recv_above"][1] * width, dtype=torch.float64)
        bufs.append(("recv_above", b))
        reqs.append(dist.irecv(b, rank + 1))
    if rank > 0:
        reqs.append(dist.isend(t[rng("send_down")].clone(), rank - 1))
        b = torch.empty(part_halo["recv_below"][1] * width, dtype=torch.float64)
        bufs.append(("recv_below", b))
        reqs.append(dist.irecv(b, rank - 1))
    for r in reqs:
        r.wait()
    for key, b in bufs:
        vec[rng(key)] = b.numpy()


def _worker(rank, size, port, nx, ny, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        part = StripPartition((0.0, 0.0), (2.0, 1.0), nx, ny, rank, size, coarsest=2)
        dm = part.dofmap
        s = fo.Space(part.mesh.coords, part.mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
        # global reference on every rank (small)
        gm = rectangle_mesh((0.0, 0.0), (2.0, 1.0), nx, ny)
        gdm = TaylorHoodDofMap(gm)
        gs = fo.Space(gm.coords, gm.cells, gdm.p2_dofmap, gdm.p1_dofmap)
        # 1. owned rows of the locally assembled operators equal the global rows
        A2 = (s.mass_p2() + s.stiffness_p2()).tocsr()
        G2 = (gs.mass_p2() + gs.stiffness_p2()).tocsr()
        own2 = np.nonzero(part.p2_owned)[0]
        sub = G2[part.p2_global[own2]][:, part.p2_global]
        assert abs(A2[own2] - sub).max() < 1e-13
        assert abs(G2[part.p2_global[own2]]).sum() == pytest.approx(abs(sub).sum())   # no coupling outside
        # 2. distributed SpMV with interleaved 2-component vectors (P2 halo, width 2)
        rng = np.random.default_rng(3)
        xg = rng.standard_normal(2 * gdm.n_p2)
        x = xg.reshape(-1, 2)[part.p2_global].ravel().copy()
        x.reshape(-1, 2)[~part.p2_owned] = 0.0                      # stale ghosts
        _exchange(dist, torch, part.p2_halo, x, 2, rank, size)
        assert np.array_equal(x.reshape(-1, 2), xg.reshape(-1, 2)[part.p2_global])
        y = fo.sp.kron(A2, fo.sp.identity(2)) @ x
        yg = fo.sp.kron(G2, fo.sp.identity(2)) @ xg
        err = abs(y.reshape(-1, 2)[own2] - yg.reshape(-1, 2)[part.p2_global[own2]]).max()
        assert err < 1e-12
        # 3. Jacobi-PCG on the P1 problem (M + K) p = b with halo exchange + all-reduce
        A1 = (s.mass_p1() + s.stiffness_p1()).tocsr()
        G1 = (gs.mass_p1() + gs.stiffness_p1()).tocsr()
        owned = part.p1_owned
        bg = rng.standard_normal(gdm.n_p1)
        b = np.where(owned, bg[part.p1_global], 0.0)
        dinv = 1.0 / A1.diagonal()

        def dot(a, c):
            t = torch.tensor([float(a[owned] @ c[owned])], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t[0])

        def matvec(v):
            _exchange(dist, torch, part.p1_halo, v, 1, rank, size)
            w = A1 @ v
            w[~owned] = 0.0
            return w
        xk = np.zeros(dm.n_p1)
        r = b.copy()
        z = dinv * r
        z[~owned] = 0.0
        p = z.copy()
        rz = dot(r, z)
        for it in range(500):
            q = matvec(p)
            alpha = rz / dot(p, q)
            xk += alpha * p
            r -= alpha * q
            if np.sqrt(dot(r, r)) < 1e-12:
                break
            z = dinv * r
            z[~owned] = 0.0
            rz_new = dot(r, z)
            p = z + (rz_new / rz) * p
            rz = rz_new
        ref = fo.spla.spsolve(G1.tocsc(), bg)
        assert abs(xk[owned] - ref[part.p1_global[owned]]).max() < 1e-9
        # ghosts of the solution are consistent copies of the owners' values
        assert abs(xk - ref[part.p1_global]).max() < 1e-9
        # 4. multigrid level bookkeeping: local Galerkin = local rediscretisation on owned rows
        lev, (rp, col, val) = part.levels[0]
        P = fo.sp.csr_matrix((val, col, rp), shape=(dm.n_p1, lev.n_p1))
        cs = fo.Space(lev.mesh.coords, lev.mesh.cells, np.zeros((lev.mesh.num_cells(), 6), int),
                      lev.mesh.cells)
        Kc = cs.stiffness_p1()
        # interior owned coarse nodes whose fine-level support lies inside the local fine mesh
        RAP = (P.T @ s.stiffness_p1() @ P).tocsr()
        inner = np.nonzero(lev.p1_ghost == 0)[0]
        inner = inner[(inner // lev.w1 > 0) & (inner // lev.w1 < lev.own_rows)]
        assert abs(RAP[inner] - Kc[inner]).max() < 1e-12
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_halo_and_cg(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 6, 8, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, "ok_%d" % r)) for r in range(2))


def test_partition_bookkeeping_three_ranks():
    nx, ny, size = 4, 12, 3
    parts = [StripPartition((0.0, 0.0), (1.0, 3.0), nx, ny, r, size, coarsest=2) for r in range(size)]
    n2 = (2 * nx + 1) * (2 * ny + 1)
    n1 = (nx + 1) * (ny + 1)
    owned2 = np.concatenate([p.p2_global[p.p2_owned] for p in parts])
    owned1 = np.concatenate([p.p1_global[p.p1_owned] for p in parts])
    assert np.array_equal(np.sort(owned2), np.arange(n2))       # every dof owned exactly once
    assert np.array_equal(np.sort(owned1), np.arange(n1))
    for a, b in zip(parts[:-1], parts[1:]):
        # what a sends up is what b receives from below (same global nodes), and vice versa
        su, rb = a.p2_halo["send_up"], b.p2_halo["recv_below"]
        assert np.array_equal(a.p2_global[su[0]: su[0] + su[1]], b.p2_global[rb[0]: rb[0] + rb[1]])
        sd, ra = b.p2_halo["send_down"], a.p2_halo["recv_above"]
        assert np.array_equal(b.p2_global[sd[0]: sd[0] + sd[1]], a.p2_global[ra[0]: ra[0] + ra[1]])
        su, rb = a.p1_halo["send_up"], b.p1_halo["recv_below"]
        assert np.array_equal(a.p1_global[su[0]: su[0] + su[1]], b.p1_global[rb[0]: rb[0] + rb[1]])
        sd, ra = b.p1_halo["send_down"], a.p1_halo["recv_above"]
        assert np.array_equal(b.p1_global[sd[0]: sd[0] + sd[1]], a.p1_global[ra[0]: ra[0] + ra[1]])
    assert all((p.p2_ghost[~p.p2_owned] == GHOST).all() for p in parts)
    assert [len(p.levels) for p in parts] == [1, 1, 1]
    # local coordinates are slices of the global lattice
    gm = rectangle_mesh((0.0, 0.0), (1.0, 3.0), nx, ny)
    for p in parts:
        assert np.array_equal(p.mesh.coords, gm.coords[p.p1_global])


def _exchange_periodic(dist, torch, halo, vec, width, rank, size):
    """The wrap-around exchange of a periodic partition in the order RcclComm::exchange issues it:
    sends (up, then down), receives (from below, then from above).  With two ranks both
    neighbours are the same peer and messages between a pair are matched in issue order."""
    t = torch.from_numpy(vec)
    up, down = (rank + 1) % size, (rank - 1) % size

    def rng(key):
        off, cnt = halo[key]
        return slice(off * width, (off + cnt) * width)
    sends = [dist.isend(t[rng("send_up")].clone(), up), dist.isend(t[rng("send_down")].clone(), down)]
    below = torch.empty(halo["recv_below"][1] * width, dtype=torch.float64)
    above = torch.empty(halo["recv_above"][1] * width, dtype=torch.float64)
    recvs = [dist.irecv(below, down), dist.irecv(above, up)]
    for r in sends + recvs:
        r.wait()
    vec[rng("recv_below")] = below.numpy()
    vec[rng("recv_above")] = above.numpy()


def _periodic_worker(rank, size, port, n, out_dir):
    import torch
    import torch.distributed as dist
    from partition import PeriodicSlabPartition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        part = PeriodicSlabPartition((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), n, n, n, rank, size, coarsest=2)
        dm = part.dofmap
        # a smooth triple-periodic field, defined by coordinates: owners set it, ghosts start stale
        f = lambda X: np.sin(2 * np.pi * X[:, 0]) * np.cos(2 * np.pi * X[:, 1]) + np.cos(2 * np.pi * X[:, 2])
        for coords, owned, halo, width in ((dm.p2_coords, part.p2_owned, part.p2_halo, 3),
                                           (dm.p1_coords, part.p1_owned, part.p1_halo, 1)):
            exact = np.repeat(f(coords)[:, None], width, axis=1) * (1.0 + np.arange(width))[None, :]
            v = np.where(owned[:, None], exact, -777.0).ravel().copy()
            _exchange_periodic(dist, torch, halo, v, width, rank, size)
            assert np.abs(v.reshape(-1, width) - exact).max() < 1e-13       # ghosts = owners' values
        # distributed mass-matrix product on the periodic P1 space: owned rows of the local operator
        # (own layers + ghost layer, x / y identified) act like the triple-periodic global operator:
        # M 1 summed over the owned rows of all ranks is the volume of the box
        s = fo.Space(part.mesh.coords, part.mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
        vol = torch.tensor([float((s.mass_p1() @ np.ones(dm.n_p1))[part.p1_owned].sum())], dtype=torch.float64)
        dist.all_reduce(vol)
        assert abs(float(vol[0]) - 1.0) < 1e-12
        # coarse level: same exchange rule on its own halo ranges
        lev = part.levels[0][0]
        cx = np.zeros((lev.n_p1, 3))
        cx[lev.vertex_dof] = lev.mesh.coords
        masters = (lev.mesh.coords[:, 0] < 1.0 - 1e-12) & (lev.mesh.coords[:, 1] < 1.0 - 1e-12)
        cx[lev.vertex_dof[masters]] = lev.mesh.coords[masters]
        exact = f(cx)
        v = np.where(lev.p1_ghost == 0, exact, -777.0)
        _exchange_periodic(dist, torch, lev.p1_halo, v, 1, rank, size)
        assert np.abs(v - exact).max() < 1e-13
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_periodic_wraparound_exchange(tmp_path):
    """PeriodicSlabPartition on two real processes: the wrap-around halo exchange in the issue
    order of RcclComm::exchange (both neighbours are the same peer) fills every ghost plane with
    its owner's values, on the fine P2 / P1 spaces and on a coarse level."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_periodic_worker, args=(2, port, 4, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, "ok_%d" % r)) for r in range(2))


# ---------------------------------------------------------------------------------------------
# unstructured partitions (recursive coordinate bisection, index-list halos) on real gloo ranks
# ---------------------------------------------------------------------------------------------
def _exchange_lists(dist, torch, lists, vec, width, add=False):
    """forward (owners -> ghosts) or, add=True, reverse (ghost copies added at the owners) exchange
    through per-neighbour index lists -- the message pattern of RcclComm::exchange_lists"""
    v2 = vec.reshape(-1, width)
    out_ptr, out_idx = (lists["recv_ptr"], lists["recv_idx"]) if add else (lists["send_ptr"], lists["send_idx"])
    in_ptr, in_idx = (lists["send_ptr"], lists["send_idx"]) if add else (lists["recv_ptr"], lists["recv_idx"])
    reqs, bufs = [], []
    for k, q in enumerate(lists["neighbour"]):
        so = out_idx[out_ptr[k]:out_ptr[k + 1]]
        ri = in_idx[in_ptr[k]:in_ptr[k + 1]]
        if so.size:
            reqs.append(dist.isend(torch.from_numpy(v2[so].copy()), int(q)))
        if ri.size:
            b = torch.empty((ri.size, width), dtype=torch.float64)
            bufs.append((ri, b))
            reqs.append(dist.irecv(b, int(q)))
    for r in reqs:
        r.wait()
    for ri, b in bufs:          # one neighbour after the other: a node may be a ghost on several ranks
        if add:
            np.add.at(v2, ri, b.numpy())
        else:
            v2[ri] = b.numpy()


def _graph_worker(rank, size, port, out_dir):
    import torch
    import torch.distributed as dist
    import grid_generator as gg
    from partition import GraphPartition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        gm, marks = gg.dfg_channel(2, 1)
        gdm = TaylorHoodDofMap(gm)
        part = GraphPartition(gm, rank, size, marks)
        dm = part.dofmap
        s = fo.Space(part.mesh.coords, part.mesh.cells, dm.p2_dofmap, dm.p1_dofmap)
        gs = fo.Space(gm.coords, gm.cells, gdm.p2_dofmap, gdm.p1_dofmap)
        g2 = part.p2_global(gdm)
        # 1. owned rows of the locally assembled operators are the global rows
        A2 = (s.mass_p2() + s.stiffness_p2()).tocsr()
        G2 = (gs.mass_p2() + gs.stiffness_p2()).tocsr()
        own2 = np.nonzero(part.p2_owned)[0]
        assert abs(A2[own2] - G2[g2[own2]][:, g2]).max() < 1e-12
        assert abs(G2[g2[own2]]).sum() == pytest.approx(abs(G2[g2[own2]][:, g2]).sum())
        # 2. forward exchange of a 2-component vector, distributed product
        rng = np.random.default_rng(11)
        xg = rng.standard_normal(2 * gdm.n_p2)
        x = xg.reshape(-1, 2)[g2].ravel().copy()
        x.reshape(-1, 2)[~part.p2_owned] = np.nan
        _exchange_lists(dist, torch, part.p2_lists, x, 2)
        assert np.array_equal(x.reshape(-1, 2), xg.reshape(-1, 2)[g2])
        y = (fo.sp.kron(A2, fo.sp.identity(2)) @ x).reshape(-1, 2)
        yg = (fo.sp.kron(G2, fo.sp.identity(2)) @ xg).reshape(-1, 2)
        assert abs(y[own2] - yg[g2[own2]]).max() < 1e-11
        # 3. additive parts of the algebraic Schur Laplacian: sum over the ranks = D diag(M)^-1 D^T,
        #    applied with forward exchange -> local rows -> reverse add
        D = s.divergence().tocsr()
        Dg = gs.divergence().tocsr()
        w = np.repeat(np.where(part.p2_owned, 1.0 / s.mass_p2().diagonal(), 0.0), 2)
        Ar = (D @ fo.sp.diags(w) @ D.T).tocsr()
        Ag = (Dg @ fo.sp.diags(np.repeat(1.0 / gs.mass_p2().diagonal(), 2)) @ Dg.T).tocsr()
        pg = rng.standard_normal(gdm.n_p1)
        p = np.where(part.p1_owned, pg[part.p1_global], np.nan)
        _exchange_lists(dist, torch, part.p1_lists, p, 1)
        t = Ar @ p
        _exchange_lists(dist, torch, part.p1_lists, t, 1, add=True)
        ref = Ag @ pg
        assert abs(t[part.p1_owned] - ref[part.p1_global[part.p1_owned]]).max() < 1e-9 * abs(ref).max()
        # 4. all-reduced dot product over owned entries = global dot product
        tt = torch.tensor([float(p[part.p1_owned] @ p[part.p1_owned])], dtype=torch.float64)
        dist.all_reduce(tt)
        assert float(tt[0]) == pytest.approx(float(pg @ pg), rel=1e-13)
        open(os.path.join(out_dir, "gok_%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [2, 3])
def test_gloo_ranks_on_a_bisected_unstructured_mesh(size, tmp_path):
    """partition.GraphPartition on real gloo ranks (world_size 2 and 3: a rank with two neighbours):
    index-list halo exchange, distributed operator application on the DFG channel mesh, and the
    additive form of the algebraic Schur Laplacian with the reverse (add) exchange -- the
    communication pattern of RcclComm::exchange_lists, checked against the undistributed oracle."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_graph_worker, args=(size, port, str(tmp_path)), nprocs=size, join=True)
    assert all(os.path.exists(os.path.join(tmp_path, "gok_%d" % r)) for r in range(size))
