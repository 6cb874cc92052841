"""GPU: the library keeps no process-wide settings (ABI 6, SURVEY.md 8b 'no global state; re-entrant').  Two "trainers" in one
process -- different model sizes, different image sizes, different per-call flags (one culls its tile lists and sorts inside the blend, the
other keeps every instance of the reference's rectangles, sorts in launches of its own and ranks with the ballot fallback), each on its own
stream with its own placement buffer -- are run INTERLEAVED, forward and backward, without a synchronisation in between, and each must
produce exactly what it produces alone: images, radii, transmittance and tile lists bit for bit (a wrong flag would change the lists), the
gradients to float-atomics noise."""
import numpy as np
import pytest
import torch

from tests.common import make_inputs, seeded_grads
from tests.test_gpu_parity import hip_forward, hip_backward, oracle_forward

pytestmark = pytest.mark.gpu


def _run(trainer, dev):
    """one forward + backward of a trainer on ITS stream with ITS flags -> everything comparable, still on the device"""
    from egogaussian_amd import _C
    d, flags, stream, grads = trainer
    with torch.cuda.stream(stream):
        g, out = hip_forward(d, dev, debug=flags)
        cap = _C.stats["capacity"]
        hb = hip_backward(g, out, grads, dev, debug=flags)
    return out, cap, hb


def _snapshot(res, P, W, H):
    from egogaussian_amd import _C
    out, cap, hb = res
    iv = _C.image_views(out[7], W, H)
    bv = _C.binning_views(out[6], P, out[0], W, H, cap)
    n = int(iv["ranges"][:, 1].max().item())
    return dict(color=out[1].cpu().numpy(), depth=out[2].cpu().numpy(), alpha=out[3].cpu().numpy(), radii=out[4].cpu().numpy(), R=out[0],
                final_T=iv["final_T"].cpu().numpy(), n_contrib=iv["n_contrib"].cpu().numpy(), ranges=iv["ranges"].cpu().numpy(),
                point_list=bv["point_list"].cpu().numpy()[:n].copy(), grads=[None if t is None else t.cpu().numpy() for t in hb])


def test_two_interleaved_trainers_with_different_flags_equal_each_alone():
    from egogaussian_amd import _C
    dev = torch.device("cuda:0")
    A = (make_inputs(20000, 96, 128, 3, 1, "sh_sr", scale_mul=2.0), 0, torch.cuda.Stream(), seeded_grads(96, 128, 5))
    B = (make_inputs(30000, 160, 200, 4, 0, "col_cov", scale_mul=2.5), _C.CALL_KEEP_ALL_INSTANCES | _C.CALL_SEPARATE_SORT | _C.CALL_BALLOT_RANK | _C.CALL_SEPARATE_COUNT,
         torch.cuda.Stream(), seeded_grads(160, 200, 6))
    dims = {id(A): (20000, 128, 96), id(B): (30000, 200, 160)}
    torch.cuda.synchronize()
    alone = {}
    for t in (A, B):
        for _ in range(2):                                           # (second pass: the placement buffer holds the first one's costs, as in the interleaved run)
            res = _run(t, dev)
            torch.cuda.synchronize()
        alone[id(t)] = _snapshot(res, *dims[id(t)])
    # B keeps every instance: its lists are the reference algorithm's; A culls: strictly shorter lists
    oB = oracle_forward(B[0])[1]
    assert np.array_equal(alone[id(B)]["point_list"].view(np.uint32), oB["point_list"]) and np.array_equal(alone[id(B)]["ranges"].view(np.uint32), oB["ranges"])
    oA = oracle_forward(A[0])[1]
    assert alone[id(A)]["point_list"].size < oA["point_list"].size
    # interleaved, no synchronisation between the calls: A fwd+bwd, B fwd+bwd, B, A, A, B ...
    last = {}
    for t in (A, B, B, A, A, B, A, B):
        last[id(t)] = _run(t, dev)
    torch.cuda.synchronize()
    for t, name in ((A, "A (culling, sort in blend)"), (B, "B (all instances, separate count + sort, ballot rank)")):
        got, ref = _snapshot(last[id(t)], *dims[id(t)]), alone[id(t)]
        for k in ("color", "depth", "alpha", "radii", "final_T", "n_contrib", "ranges", "point_list"):
            assert np.array_equal(got[k], ref[k]), f"trainer {name}: {k} differs between the interleaved run and the run alone"
        assert got["R"] == ref["R"]
        for a, b in zip(got["grads"], ref["grads"]):
            if a is None or a.size == 0:
                continue
            assert np.abs(a - b).max() <= 2e-6 * (np.abs(b).max() + 1e-30), f"trainer {name}: a gradient moved by more than atomics order"
