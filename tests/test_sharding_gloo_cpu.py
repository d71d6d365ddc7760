"""N > 1 host path on CPU: two gloo ranks shard the voxel columns exactly as the GPU engine does
(geobo_amd/sharding.py), exchange the partial AkA with the product's all-reduce and the mu/var slices with its
all-gather, and must reproduce the unsharded oracle posterior.  The row-sharded form of round 3 (gather_rows of AkA row blocks, one
all-reduce of partial means and sums of squares of the transposed posterior) is covered the same way."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from scipy.linalg import cholesky, solve_triangular
    from geobo_amd.sharding import allreduce_sum_, assemble_columns, gather_slices, shard_columns
    from oracle import geobo_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = load_golden("tiny_matern32.npz")
    N, N_pad, props = 480, 512, (0, 1, 2)
    P3 = O.grid_points((10, 8, 6), (100., 100., 100.))
    lengths = O.mutate_lengths(f["gp_length_in"].copy())
    W = O.weight_matrix((1.0, 0.2, 0.2))
    A = {0: f["A_g"], 1: f["A_m"]}
    sel, y = f["sel"], f["Fs3"]
    mg, md = 80, sel.size
    M = 2 * mg + md
    c0, c1 = shard_columns(N_pad, world, rank)
    cols = np.arange(c0, min(c1, N))
    # this rank's columns of AK (contraction over ALL voxels), exactly the engine's step 1
    D2 = O.sqdist(P3, P3[cols])
    AK = {}
    for j in props:
        blk = np.zeros((M, c1 - c0))
        for s_ in (0, 1):
            blk[s_ * mg:(s_ + 1) * mg, :cols.size] = A[s_] @ O.k_block("matern32", D2, lengths, W, s_, j)
        blk[2 * mg:, :cols.size] = O.k_block("matern32", D2[sel], lengths, W, 2, j)
        AK[j] = blk
    part = np.zeros((M, M))
    for s_ in (0, 1):
        part[:, s_ * mg:(s_ + 1) * mg] = AK[s_][:, :cols.size] @ A[s_][:, cols].T
    t = torch.from_numpy(part)
    allreduce_sum_(t, world)                                     # <- product collective #1
    AkA = t.numpy()
    AkA[:2 * mg, 2 * mg:] = AkA[2 * mg:, :2 * mg].T
    AkA[2 * mg:, 2 * mg:] = O.k_block("matern32", O.sqdist(P3[sel]), lengths, W, 2, 2)
    AkA += np.diag(np.r_[np.full(2 * mg, 0.01), np.full(md, 0.01)])
    L = cholesky(AkA, lower=True)
    u = solve_triangular(L, y, lower=True)
    mu_l, var_l = [], []
    for j in props:
        V = solve_triangular(L, AK[j], lower=True)
        mu_l.append(V.T @ u)
        var_l.append(1.0 - np.einsum("mq,mq->q", V, V))
    mu_parts = gather_slices(torch.from_numpy(np.concatenate(mu_l)), len(props), N_pad, world)   # <- collective #2
    var_parts = gather_slices(torch.from_numpy(np.concatenate(var_l)), len(props), N_pad, world)
    mu = assemble_columns(mu_parts, props, N, N_pad, world)
    var = assemble_columns(var_parts, props, N, N_pad, world)
    if rank == 0:
        q.put((mu, var, AkA))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_sharded_posterior_matches_reference(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    mu, var, AkA = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = load_golden("tiny_matern32.npz")
    assert np.abs(AkA - f["AkA"]).max() / np.abs(f["AkA"]).max() < 1e-13
    assert np.abs(mu - f["mu"]).max() / np.abs(f["mu"]).max() < 1e-10
    assert np.abs(var - f["var"]).max() / np.abs(f["var"]).max() < 1e-10


def _rows_worker(rank, world, port, q):
    """The ROW-sharded form (round 3; engine._assemble_rows / _aka_local_rows / _posterior_rows): a rank owns mg / world sensor rows
    of each operator and a share of the drill rows; AkA arrives by gather_rows, the partial sums of squares of the transposed
    posterior V = (L^-1 A3) K meet in ONE allreduce_sum_; the mean K (A3^T L^-T u) is formed whole on every rank."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from scipy.linalg import cholesky, solve_triangular
    from geobo_amd.sharding import allreduce_sum_, gather_rows
    from oracle import geobo_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = load_golden("tiny_matern32.npz")
    N, props = 480, (0, 1, 2)
    P3 = O.grid_points((10, 8, 6), (100., 100., 100.))
    lengths = O.mutate_lengths(f["gp_length_in"].copy())
    W = O.weight_matrix((1.0, 0.2, 0.2))
    A = {0: f["A_g"], 1: f["A_m"]}
    sel, y = f["sel"], f["Fs3"]
    mg, md = 80, sel.size
    M = 2 * mg + md
    rows_r = mg // world
    a0 = rank * rows_r
    D2 = O.sqdist(P3, P3)
    K = {(i, j): O.k_block("matern32", D2, lengths, W, i, j) for i in (0, 1, 2) for j in props}
    # this rank's sensor rows of A K over ALL voxels, every property block
    AKr = {(s_, j): A[s_][a0:a0 + rows_r] @ K[(s_, j)] for s_ in (0, 1) for j in props}
    loc = np.stack([np.concatenate([AKr[(s_, 0)] @ A[0].T, AKr[(s_, 1)] @ A[1].T], axis=1) for s_ in (0, 1)])   # (2, rows_r, 2 mg)
    allrows = gather_rows(torch.from_numpy(loc), world).numpy()                       # <- product collective #1
    AkA = np.zeros((M, M))
    for src in range(world):
        for s_ in (0, 1):
            AkA[s_ * mg + src * rows_r:s_ * mg + (src + 1) * rows_r, :2 * mg] = allrows[src, s_]
    AkA[2 * mg:, :2 * mg] = np.concatenate([K[(2, 0)][sel] @ A[0].T, K[(2, 1)][sel] @ A[1].T], axis=1)   # drill rows: every rank
    AkA[:2 * mg, 2 * mg:] = AkA[2 * mg:, :2 * mg].T
    AkA[2 * mg:, 2 * mg:] = O.k_block("matern32", O.sqdist(P3[sel]), lengths, W, 2, 2)
    AkA += np.diag(np.r_[np.full(2 * mg, 0.01), np.full(md, 0.01)])
    L = cholesky(AkA, lower=True)
    Linv = solve_triangular(L, np.eye(M), lower=True)
    u = Linv @ y
    w = Linv.T @ u
    dper = -(-md // world)
    drows = np.arange(rank * dper, min(md, (rank + 1) * dper))                        # this rank's share of the drill rows
    mine = np.r_[a0:a0 + rows_r, mg + a0:mg + a0 + rows_r, 2 * mg + drows].astype(int)
    red = np.zeros((len(props), N))
    mu = np.zeros((len(props), N))
    e_d = np.zeros(N)
    e_d[sel] = w[2 * mg:]
    for jj, j in enumerate(props):
        # the mean whole on every rank: mu_j = K_0j (A_g^T w_g) + K_1j (A_m^T w_m) + K_2j (drill weights at their voxels)
        mu[jj] = K[(0, j)].T @ (A[0].T @ w[:mg]) + K[(1, j)].T @ (A[1].T @ w[mg:2 * mg]) + K[(2, j)].T @ e_d
        Z_g, Z_m, Z_d = Linv[mine, :mg] @ A[0], Linv[mine, mg:2 * mg] @ A[1], Linv[mine, 2 * mg:]
        V = Z_g @ K[(0, j)] + Z_m @ K[(1, j)] + Z_d @ K[(2, j)][sel]
        red[jj] = np.einsum("mq,mq->q", V, V)
    t = torch.from_numpy(red)
    allreduce_sum_(t, world)                                                           # <- product collective #2
    if rank == 0:
        q.put((mu.reshape(-1).copy(), (1.0 - t).reshape(-1).numpy().copy(), AkA))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_transposed_posterior_matches_reference(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    mu, var, AkA = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = load_golden("tiny_matern32.npz")
    assert np.abs(AkA - f["AkA"]).max() / np.abs(f["AkA"]).max() < 1e-13
    assert np.abs(mu - f["mu"]).max() / np.abs(f["mu"]).max() < 1e-10
    assert np.abs(var - f["var"]).max() / np.abs(f["var"]).max() < 1e-10


def _xchg_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from geobo_amd.sharding import exchange_blocks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blk = 6
    send = torch.arange(world * blk, dtype=torch.float64).reshape(world, blk) + 1000.0 * rank   # row d -> rank d
    recv = exchange_blocks(send, world)
    # the two-phase form (asynchronous on RCCL, complete at start on gloo) delivers the same blocks
    from geobo_amd.sharding import exchange_blocks_finish, exchange_blocks_start
    recv2, work = exchange_blocks_start(send, world)
    exchange_blocks_finish(work)
    assert work is None and torch.equal(recv2, recv)
    # row blocks of AkA (row-sharded lattice Gram): all-gather, every rank ends with the same (world, rows, cols) stack
    from geobo_amd.sharding import gather_rows
    rows = gather_rows(torch.full((2, 3), float(rank), dtype=torch.float64) + torch.arange(3, dtype=torch.float64), world)
    assert rows.shape == (world, 2, 3) and all(torch.equal(rows[r], torch.full((2, 3), float(r), dtype=torch.float64) +
                                                           torch.arange(3, dtype=torch.float64)) for r in range(world))
    q.put((rank, recv.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_blocks_is_an_all_to_all():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xchg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blk = 6
    for r in range(world):
        for src in range(world):   # row src of rank r's result = row r of rank src's send buffer
            want = np.arange(r * blk, (r + 1) * blk, dtype=np.float64) + 1000.0 * src
            assert np.array_equal(got[r][src], want)
