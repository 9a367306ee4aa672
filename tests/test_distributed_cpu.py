"""CPU, world_size 2 over gloo: the multi-GPU path by construction.  The shard arithmetic on each rank is done
by the oracle (there is no GPU here); what is under test is the product's partitioning (icp_shard_range), the
moment-vector layout, the single all-reduce per iteration and the host loop (icp_host_loop_* -- the code the
device loop runs), which must give every rank the same R, t, stop decision and composed transform as one rank
holding the whole cloud."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleShard:
    """oracle-backed shard operations in the product's precision design: fp32 matching/transform arithmetic,
    fp64 moments"""

    def __init__(self, orc, P_shard, Q):
        self.orc, self.P, self.Q = orc, np.array(P_shard, dtype=np.float32), np.asarray(Q, dtype=np.float32)
        self.prev_idx = None

    def moments(self):
        mom = np.zeros(32)
        P64, Q64 = self.P.astype(np.float64), self.Q.astype(np.float64)
        if self.prev_idx is not None:                      # error of the motion applied last, old matches
            mom[0] = ((Q64[self.prev_idx] - P64) ** 2).sum()
        if len(self.P):
            idx = self.orc.nn(self.P, self.Q)
            Qi = Q64[idx]
            mom[1] = len(self.P)
            mom[2:5] = P64.sum(0)
            mom[5:8] = Qi.sum(0)
            mom[8:17] = (Qi.T @ P64).reshape(9)
            mom[17] = (P64 * P64).sum()
            mom[18] = (Qi * Qi).sum()
            self.prev_idx = idx
        return mom

    def apply(self, R, t):
        if len(self.P):
            self.P = self.orc.transform(self.P, R.astype(np.float32), t.astype(np.float32))


def _worker(rank, world, port, n_pts, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib
    from __graft_entry__ import load_package
    pkg = load_package()
    orc = oracle_lib.Oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W = int(round(n_pts ** 0.5))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:n_pts]
    M = pkg.datasets.make_model_gpu(pkg.datasets.synthetic_grid(W, np.float32), *pkg.datasets.P2P_GPU)
    Ps, begin = pkg.distributed.shard(D, rank, world)

    def allreduce(vec):
        tvec = torch.from_numpy(vec)
        dist.all_reduce(tvec)              # in place, SUM: the one collective of an iteration

    ops = OracleShard(orc, Ps, M)
    hl = pkg.distributed.HostLoop(pkg.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6, precision=pkg.ICP_F32)
    st = pkg.distributed.drive(ops, hl, allreduce)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), T=st["T"], err=st["err"], iterations=st["iterations"], begin=begin,
             moved=ops.P, idx=ops.prev_idx if ops.prev_idx is not None else np.zeros(0, np.int32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pts", [32 * 32, 31 * 31 - 5])   # even split and a ragged one
def test_two_ranks_equal_one_rank(tmp_path, pkg, orc, n_pts):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n_pts, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank reaches the same decision and the same transform, bit for bit
    assert int(r0["iterations"]) == int(r1["iterations"])
    assert np.array_equal(r0["T"], r1["T"]) and np.array_equal(r0["err"], r1["err"])
    assert int(r0["begin"]) == 0 and int(r1["begin"]) == len(r0["moved"])

    # one rank holding the whole cloud: same driver, no communication
    W = int(round(n_pts ** 0.5))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:n_pts]
    M = pkg.datasets.make_model_gpu(pkg.datasets.synthetic_grid(W, np.float32), *pkg.datasets.P2P_GPU)
    ops = OracleShard(orc, D, M)
    hl = pkg.distributed.HostLoop(pkg.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6, precision=pkg.ICP_F32)
    one = pkg.distributed.drive(ops, hl, lambda v: None)
    assert one["iterations"] == int(r0["iterations"])
    assert np.abs(one["T"] - r0["T"]).max() < 1e-9 and np.abs(one["err"] - r0["err"]).max() < 1e-9
    assert np.array_equal(np.concatenate([r0["idx"], r1["idx"]]), ops.prev_idx)   # global idx = concatenation

    # and the oracle's own loop (fp32 matching, fp64 minimisation) agrees
    want = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert want["iterations"] == one["iterations"]
    assert np.abs(want["T"] - one["T"]).max() < 1e-9
    assert np.abs(want["err"] - one["err"]).max() < 1e-9


def test_host_loop_stop_rule_and_limits(pkg):
    """the reference's exit conditions (src/ICP_CPU.c:267-269) on synthetic moment vectors"""
    HL = pkg.distributed.HostLoop

    def mom(err_sumsq, n=100.0):
        v = np.zeros(32)
        v[0], v[1] = err_sumsq, n
        v[8], v[12], v[16] = 1.0, 1.0, 1.0     # cross-covariance = identity -> R = I
        return v

    hl = HL(max_iter=5, tol=1e-3)
    done, R, t = hl.advance(mom(0.0))
    assert not done and np.allclose(R, np.eye(3))
    hl.note_applied()
    done, _, _ = hl.advance(mom(100.0 * 0.5 ** 2))          # E[1] = 0.5
    assert not done
    hl.note_applied()
    done, _, _ = hl.advance(mom(100.0 * 0.4995 ** 2))       # |dE| = 5e-4 < tol -> stop, counter not incremented
    st = hl.state()
    assert done and st["iterations"] == 1 and st["passes"] == 2 and abs(st["err"][2] - 0.4995) < 1e-12

    hl = HL(max_iter=3, tol=0.0, fixed_iterations=True)      # runs exactly max_iter passes
    k = 0
    done, _, _ = hl.advance(mom(0.0))
    while not done:
        hl.note_applied()
        k += 1
        done, _, _ = hl.advance(mom(1.0))
    assert k == 3 and hl.state()["iterations"] == 3
    with pytest.raises(pkg.IcpError):
        HL(max_iter=0)


# ---------------------------------------------------------------------------------------------------
# the host-memory communicator (icp_lcomm_*): what the single-node resident loop exchanges its vector through
# ---------------------------------------------------------------------------------------------------
def _lcomm_worker(rank, world, id_hex, n_pts, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from __graft_entry__ import load_package
    pkg = load_package()
    orc = oracle_lib.Oracle()
    W = int(round(n_pts ** 0.5))
    D = pkg.datasets.synthetic_grid(W, np.float32)[:n_pts]
    M = pkg.datasets.make_model_gpu(pkg.datasets.synthetic_grid(W, np.float32), *pkg.datasets.P2P_GPU)
    Ps, begin = pkg.distributed.shard(D, rank, world)
    with pkg.LocalComm(bytes.fromhex(id_hex), rank, world) as comm:
        # raw exchanges first: rank-dependent vectors, several rounds back to back (sequence parity, no overwrites)
        sums = [comm.allreduce(np.arange(32, dtype=np.float64) * (rank + 1) + k) for k in range(50)]

        def allreduce(vec):
            vec[:] = comm.allreduce(vec)

        ops = OracleShard(orc, Ps, M)
        hl = pkg.distributed.HostLoop(pkg.ICP_POINT_TO_POINT, max_iter=40, tol=1e-6, precision=pkg.ICP_F32)
        st = pkg.distributed.drive(ops, hl, allreduce)
    np.savez(os.path.join(out_dir, f"lrank{rank}.npz"), T=st["T"], err=st["err"], iterations=st["iterations"], sums=np.array(sums))


def test_local_communicator_three_ranks(tmp_path, pkg, orc):
    import torch.multiprocessing as mp
    world, n_pts = 3, 31 * 31 - 5
    id_hex = pkg.Context.comm_random_id().hex()
    mp.spawn(_lcomm_worker, args=(world, id_hex, n_pts, str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"lrank{k}.npz") for k in range(world)]
    want = np.array([sum(np.arange(32, dtype=np.float64) * (q + 1) + k for q in range(world)) for k in range(50)])
    for k in range(world):
        assert np.array_equal(r[k]["sums"], want)
        assert np.array_equal(r[k]["T"], r[0]["T"]) and np.array_equal(r[k]["err"], r[0]["err"])   # bit-identical on every rank
    D = pkg.datasets.synthetic_grid(31, np.float32)[:n_pts]
    M = pkg.datasets.make_model_gpu(pkg.datasets.synthetic_grid(31, np.float32), *pkg.datasets.P2P_GPU)
    ref = orc.icp_p2p_f32x(D, M, 40, 1e-6)
    assert int(r[0]["iterations"]) == ref["iterations"] and np.abs(ref["T"] - r[0]["T"]).max() < 1e-9


def test_dealt_shards_are_a_partition_of_compact_blocks(pkg):
    """distributed.curve_order / shard_cyclic_index (what bench.py --config s5 --gpus N deals the moving cloud by): the order is a Hilbert
    curve (on a full lattice every step goes to a face neighbour), the ranks' shares partition the cloud whatever the sizes, a block
    is a contiguous stretch of the curve, and no rank gets the same corner of every cell (the digit-sum dealing)"""
    d = pkg.distributed
    b = 4
    g = np.arange(1 << b)
    A = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    o = d.curve_order(A, bits=b)
    assert np.array_equal(np.sort(o), np.arange(len(A)))
    assert (np.abs(np.diff(A[o], axis=0)).sum(1) == 1).all()
    for n, world, block in ((10_007, 8, 128), (4096, 3, 1024), (100, 8, 16384), (65_536, 1, 512)):
        order = np.random.default_rng(n).permutation(n)
        shares = [d.shard_cyclic_index(n, r, world, block, order) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(shares)), np.arange(n))
        pos = np.empty(n, dtype=np.int64); pos[order] = np.arange(n)
        for s in shares:                     # every share: whole blocks of the order (the last one may be short), ascending
            p = pos[s]
            assert (np.diff(p) > 0).all() and all((p[k] // block == p[k + 1] // block) or (p[k + 1] % block == 0) for k in range(len(p) - 1))
    owners = np.array([[r for r in range(8) if bidx * 64 in set(d.shard_cyclic_index(64 * 512, r, 8, 64).tolist())][0] for bidx in range(512)])
    for level in (1, 8, 64):                 # the eight children of a cell of any level go to eight different ranks ...
        kids = owners.reshape(-1, 8, level)[:, :, 0]
        assert all(len(set(row.tolist())) == 8 for row in kids)
    assert len(set(owners[::8].tolist())) == 8   # ... and the FIRST child of the cells of a level does not always go to the same one
