"""Worker for tests/test_domain_cpu.py (gloo, CPU only): the host side of the slab decomposition --
slab geometry, ownership, the neighbour ring exchange (including the two-rank case where both
neighbours are one peer) and the halo selection rule -- checked by computing every owned particle's
force from owned + received halo copies with the oracle and comparing with the global oracle result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from moleculardynamics.jl_amd.domain import Exchanger, slab_bounds, owner_of, halo_selection, neighbours
    from oracle import oracle as orc
    from tests.util import lj_system

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ex = Exchanger()
    assert not ex.on_device and (ex.left, ex.right) == neighbours(rank, world)

    # 1. ring exchange: payloads identify (sender, direction); for world == 2 both messages come from one peer
    sl = torch.full((3,), 10.0 * rank + 1, dtype=torch.float64)   # to the left
    sr = torch.full((5,), 10.0 * rank + 2, dtype=torch.float64)   # to the right
    nfl, nfr = ex.all_counts((3, 5))
    assert (nfl, nfr) == (5, 3)       # from the left I get what it sent right (5), from the right what it sent left (3)
    rl, rr = torch.empty(nfl, dtype=torch.float64), torch.empty(nfr, dtype=torch.float64)
    ex.sendrecv(sl, sr, rl, rr)
    assert torch.all(rl == 10.0 * ex.left + 2) and torch.all(rr == 10.0 * ex.right + 1)
    assert ex.allreduce([1.0, float(rank)]) == [float(world), float(sum(range(world)))]
    assert ex.allreduce([float(rank)], op="max") == [float(world - 1)]

    # 2. decomposition geometry against the oracle
    n, rc = 3000, 2.5
    s = lj_system(n, permute=5)
    L = s["box"][0]
    pot = orc.make_pot(orc.POT_LJ, [1.0, 1.0, 2.5])
    f_ref, u_ref, w_ref, _ = orc.forces_cells(s["x"], s["box"], rc, pot, s["diam"], nthreads=1)
    own = owner_of(s["x"][:, 0], L, world)
    assert np.array_equal(np.bincount(own, minlength=world).sum(), n)
    lo, hi = slab_bounds(L, world, rank)
    mine = np.nonzero(own == rank)[0]
    assert np.all((s["x"][mine, 0] >= lo) & (s["x"][mine, 0] < hi + 1e-12))
    xm = s["x"][mine]
    to_l, to_r = halo_selection(xm[:, 0], lo, hi, rc)
    shift_l = L if rank == 0 else 0.0           # crossing the global periodic face
    shift_r = -L if rank == world - 1 else 0.0

    def pack(mask, shift):
        rec = np.concatenate([xm[mask] + np.array([shift, 0.0, 0.0]), mine[mask, None].astype(np.float64)], axis=1)
        return torch.from_numpy(np.ascontiguousarray(rec).ravel())

    bl, br = pack(to_l, shift_l), pack(to_r, shift_r)
    nfl, nfr = ex.all_counts((bl.numel(), br.numel()))
    rl, rr = torch.empty(nfl, dtype=torch.float64), torch.empty(nfr, dtype=torch.float64)
    ex.sendrecv(bl, br, rl, rr)
    halo = np.concatenate([rl.numpy().reshape(-1, 4), rr.numpy().reshape(-1, 4)], axis=0)
    # forces on my particles from owned + halo copies; x is NOT periodic locally (the copies are translated),
    # y and z are: give the oracle a box three times as long in x
    allx = np.concatenate([xm, halo[:, :3]], axis=0)
    big = np.array([3.0 * L, L, L])
    f_loc, _, _, _ = orc.forces_brute(allx + np.array([L, 0, 0]), big, rc, pot, np.ones(len(allx)))
    err = np.abs(f_loc[: len(mine)] - f_ref[mine]).max() / max(1.0, np.abs(f_ref).max())
    assert err <= 1e-11, f"rank {rank}: force from owned+halo differs from the global force: {err:.2e}"
    # every halo copy really is another rank's particle
    assert not np.intersect1d(halo[:, 3].astype(np.int64), mine).size or world == 1
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok: own={len(mine)} halo={len(halo)} err={err:.1e}")


if __name__ == "__main__":
    main()
