"""Two host threads, one lf_ctx each, calling lf_match_node_pair at the same time (the shape of the reference's QThreadPool
fan-out over candidate nodes, src/graph_manager.cpp:555, with the per-thread context INTEGRATION.md prescribes): every result
must equal the one the same call returns when nothing else runs."""
import threading

import numpy as np
import pytest

from lineslam_amd import capi, synth

pytestmark = pytest.mark.gpu


def _bytes(r):
    return (np.array(list(r.T), np.float32).tobytes(), int(r.valid), int(r.n_matches), int(r.n_inliers), float(r.rmse),
            int(r.ransac_best_iter), int(r.refine_rounds))


def test_two_threads_two_contexts_match_serial(built_lib):
    g, d, _ = synth.sequence(4, seed=11)
    P = capi.default_params(launch=True)
    front = capi.Context(640, 480, max_batch=2, params=P)
    recs = [front.detect3d(g[k], d[k], synth.K_TUM, frame_id=k) for k in range(4)]
    front.close()
    jobs = {0: [(1, 0), (2, 1), (3, 2)], 1: [(3, 2), (2, 0), (3, 1)]}      # (newer, older) per thread
    ctxs = [capi.Context(640, 480, max_batch=2, params=P) for _ in range(2)]
    serial = {t: [_bytes(ctxs[t].match_node_pair(recs[a], a, recs[b], b, allow_overflow=True)) for a, b in jobs[t]] for t in (0, 1)}
    assert any(s[1] for s in serial[0] + serial[1]), "the pairs of this test must produce valid transforms"
    got, errs = {0: [], 1: []}, []
    start = threading.Barrier(2)

    def run(t):
        try:
            start.wait()
            for rep in range(12):
                for a, b in jobs[t]:
                    got[t].append(_bytes(ctxs[t].match_node_pair(recs[a], a, recs[b], b, allow_overflow=True)))
        except Exception as e:                     # (a LinefrontError of one thread fails the test, not only that thread)
            errs.append(repr(e))
    th = [threading.Thread(target=run, args=(t,)) for t in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in (0, 1):
        assert len(got[t]) == 12 * len(jobs[t])
        for k, r in enumerate(got[t]):
            assert r == serial[t][k % len(jobs[t])], "thread %d, call %d differs from the serial result" % (t, k)
    for c in ctxs:
        c.close()


def test_create_failure_message_survives(built_lib):
    """lf_ctx_create_caps that cannot get its device memory: LF_ERR_CAPACITY, and lf_last_error(NULL) still names the buffer
    although the context is gone (linefront.h)."""
    with pytest.raises(capi.LinefrontError) as ei:
        capi.Context(640, 480, max_batch=400000)          # ~ 8 TB of batch buffers
    assert ei.value.status == capi.LF_ERR_CAPACITY
    assert "device memory" in str(ei.value) and "needs" in str(ei.value)
