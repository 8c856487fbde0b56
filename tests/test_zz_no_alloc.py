"""GPU: the allocation-free matcher entry point orbamd_knn2_dev_noalloc, called through the C ABI
(ctypes), against orbamd_knn2_dev on the same inputs -- bit for bit, on two calls in a row, the
second with a smaller input (the library's scratch array still holds the first call's partial keys)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from visual_slam_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _call(ctx, fn, q, t):
    L = _lib.lib()
    nq, nt = q.shape[0], t.shape[0]
    idx = torch.full((max(nq, 1), 2), -9, dtype=torch.int32, device=q.device)
    dist = torch.full((max(nq, 1), 2), -9, dtype=torch.int32, device=q.device)
    good = torch.full((max(nq, 1),), 9, dtype=torch.uint8, device=q.device)
    vp = ctypes.c_void_p
    st = getattr(L, fn)(ctx._h, vp(q.data_ptr()), nq, vp(t.data_ptr() if nt else 0), nt, vp(idx.data_ptr()),
                        vp(dist.data_ptr()), vp(good.data_ptr()))
    assert st == _lib.OK, (fn, st)
    ctx.sync()
    return idx[:nq].cpu().numpy(), dist[:nq].cpu().numpy(), good[:nq].cpu().numpy()


@pytest.mark.parametrize("matcher", [_lib.MATCHER_MATRIX, _lib.MATCHER_VALU])
def test_noalloc_matcher_equals_allocating_entry_point(matcher):
    ctx = _lib.Context(0)
    try:
        ctx.set_matcher(matcher)
        dev = torch.device("cuda", 0)
        # (nq, nt): a many-way train split, then a smaller call that leaves stale partial keys behind
        # its own, then one without a split, then fewer than two train rows
        # then two large query sets for which the split count has to be lowered until the partial keys fit the
        # fixed 4 MiB array (the largest one is the entry point's limit)
        cases = [(3000, 5000, 1), (700, 900, 2), (300, 200, 3), (65, 1, 4), (33000, 8192, 5)]
        if matcher == _lib.MATCHER_MATRIX:
            cases.append((_lib.NOALLOC_MAX_QUERIES, 4096, 6))
        for nq, nt, seed in cases:
            qh = synth.synth_descriptors(nq, 100 + seed)
            th = synth.synth_descriptors(nt, 200 + seed)
            th[: min(nt, 40)] = qh[: min(nt, 40)]              # exact hits and ties
            q, t = torch.from_numpy(qh).to(dev), torch.from_numpy(th).to(dev)
            got = _call(ctx, "orbamd_knn2_dev_noalloc", q, t)
            want = _call(ctx, "orbamd_knn2_dev", q, t)
            for g, w in zip(got, want):
                assert np.array_equal(g, w)
            if nq <= 5000:
                eidx, edist = oracle.c.knn2(qh, th)
                assert np.array_equal(got[0], eidx) and np.array_equal(got[1], edist)
    finally:
        ctx.close()


def test_noalloc_matcher_leaves_device_memory_alone():
    """The advertised property itself: on a context that has never matched anything, with timing switched on
    (the allocating entry points then create events), a call changes neither the free device memory nor
    what torch has reserved."""
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(synth.synth_descriptors(6000, 7)).to(dev)
    t = torch.from_numpy(synth.synth_descriptors(7000, 8)).to(dev)
    idx = torch.empty((6000, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((6000, 2), dtype=torch.int32, device=dev)
    # what the HIP runtime itself sets up lazily (code objects, queues) happens on another context first
    warm = _lib.Context(0)
    warm.knn2_dev_noalloc(q.data_ptr(), 6000, t.data_ptr(), 7000, idx.data_ptr(), dist.data_ptr())
    warm.sync()
    warm.close()
    ctx = _lib.Context(0)
    try:
        ctx.timing_enable(True)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(dev)
        reserved0 = torch.cuda.memory_reserved(dev)
        vp = ctypes.c_void_p
        for _ in range(3):
            st = _lib.lib().orbamd_knn2_dev_noalloc(ctx._h, vp(q.data_ptr()), 6000, vp(t.data_ptr()), 7000,
                                                    vp(idx.data_ptr()), vp(dist.data_ptr()), vp(0))
            assert st == _lib.OK
        ctx.sync()
        free1, _ = torch.cuda.mem_get_info(dev)
        assert free1 == free0 and torch.cuda.memory_reserved(dev) == reserved0
        assert ctx.timing_read()["knn2"][1] == 0          # nothing was bracketed by events either
    finally:
        ctx.close()


def test_descriptor_pointers_must_be_16_byte_aligned():
    ctx = _lib.Context(0)
    try:
        buf = torch.zeros(4096, dtype=torch.uint8, device="cuda:0")
        out = torch.zeros(64, dtype=torch.int32, device="cuda:0")
        vp = ctypes.c_void_p
        L = _lib.lib()
        for fn in ("orbamd_knn2_dev", "orbamd_knn2_dev_noalloc"):
            st = getattr(L, fn)(ctx._h, vp(buf.data_ptr() + 4), 8, vp(buf.data_ptr() + 1024), 8, vp(out.data_ptr()),
                                vp(out.data_ptr() + 128), vp(0))
            assert st == _lib.E_BADARG
            st = getattr(L, fn)(ctx._h, vp(buf.data_ptr()), 8, vp(buf.data_ptr() + 1024 + 8), 8, vp(out.data_ptr()),
                                vp(out.data_ptr() + 128), vp(0))
            assert st == _lib.E_BADARG
    finally:
        ctx.close()


def test_noalloc_matcher_rejects_oversized_query_sets():
    ctx = _lib.Context(0)
    try:
        L = _lib.lib()
        buf = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
        vp = ctypes.c_void_p
        st = L.orbamd_knn2_dev_noalloc(ctx._h, vp(buf.data_ptr()), _lib.NOALLOC_MAX_QUERIES + 1, vp(buf.data_ptr()), 2,
                                       vp(buf.data_ptr()), vp(buf.data_ptr()), vp(0))
        assert st == _lib.E_CAPACITY
        assert b"ORBAMD_NOALLOC_MAX_QUERIES" in L.orbamd_last_error(ctx._h)
    finally:
        ctx.close()
