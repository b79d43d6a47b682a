"""Deferred join (include/nimble_amd.h, ABI minor 5): ONE handle whose slices run on internal streams and are not joined
per call - the forward pass of one slice overlaps the backward pass of another across consecutive calls, like the four-handle pattern.
The results must be BIT FOR BIT those of the joined calls: next states, status words, warm start, both gradients - with the per-slice
loss enqueued on the slices' own streams, three steps in a row (the second and third start while the first's slices may still run)."""
import numpy as np
import pytest
import torch

from util import contact_inputs

pytestmark = pytest.mark.gpu


def test_deferred_join_equals_the_joined_calls_bit_for_bit():
    import nimblephysics_amd as na
    B = 4096
    md, s, a = contact_inputs("atlas20", B, 11, joint_noise=0.02, vel_noise=0.01, action_noise=0.0)
    dev = torch.device("cuda:0")
    ref = na.World(md, device=dev)
    st = ref.to_soa(torch.tensor(s, device=dev)); at = ref.to_soa(torch.tensor(a, device=dev))
    want = []
    for step in range(3):
        ref.reset_lcp_cache()
        nxt, sv, status = ref.step_soa(st, at, want_saved=True)
        gs, ga = ref.backward_soa(sv, 2.0 * nxt)
        want.append((nxt.clone(), status.clone(), ref.lcp_cache.clone(), gs.clone(), ga.clone()))
    torch.cuda.synchronize()

    w = na.World(md, device=dev)
    w.set_deferred_join(True)
    sl = w.slices(B)
    assert len(sl) == 4 and sl[0][1] == 0 and sl[-1][2] == B and all(sl[i][2] == sl[i + 1][1] for i in range(3))
    n2, k, m = 2 * w.n, w.k, w.m
    bufs = [dict(nxt=torch.empty((n2, B), dtype=torch.float64, device=dev), saved=torch.empty(w.saved_bytes(B), dtype=torch.uint8, device=dev),
                 status=torch.empty(B, dtype=torch.int32, device=dev), cache=torch.empty((m, B), dtype=torch.float64, device=dev),
                 g=torch.empty((n2, B), dtype=torch.float64, device=dev), gs=torch.empty((n2, B), dtype=torch.float64, device=dev),
                 ga=torch.empty((k, B), dtype=torch.float64, device=dev)) for _ in range(3)]
    w.fork()                                                                                 # (the inputs were produced on this stream)
    for step in range(3):
        b = bufs[step]
        w.step_into(st, at, b["nxt"], b["saved"], b["status"], None, b["cache"])          # returns with its slices in flight
        for stream, lo, hi in sl:                                                            # the loss of a slice on the slice's stream
            with torch.cuda.stream(stream):
                torch.mul(b["nxt"][:, lo:hi], 2.0, out=b["g"][:, lo:hi])
        w.backward_into(b["saved"], b["g"], b["gs"], b["ga"])
    w.join()
    torch.cuda.synchronize()
    for step in range(3):
        b = bufs[step]
        nxt, status, cache, gs, ga = want[step]
        assert torch.equal(b["nxt"], nxt) and torch.equal(b["status"], status) and torch.equal(b["cache"], cache), step
        assert torch.equal(b["gs"], gs) and torch.equal(b["ga"], ga), step
    # back to joined calls on the same handle
    w.set_deferred_join(False)
    w.reset_lcp_cache()
    nxt, sv, status = w.step_soa(st, at, want_saved=True)
    torch.cuda.synchronize()
    assert torch.equal(nxt, want[0][0]) and torch.equal(status, want[0][1])
