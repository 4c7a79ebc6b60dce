"""GPU: lf_line_matching_node_pair (k_match: no descDiff matrix, LDS atomic minima on live entries) against the golden vectors of
the source-independent numpy restatement that DOES build the matrix (tests/golden/match_fixtures.npz)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _golden as G   # noqa: E402

pytestmark = pytest.mark.gpu


def test_hip_line_matching_equals_the_independent_vectors(built_lib):
    from lineslam_amd import capi
    ctx = capi.Context(640, 480, max_batch=2, params=capi.default_params(launch=True))
    tot = 0
    over = 0
    for k, c in enumerate(G.match_cases()):
        if len(c["mq"]) > ctx.caps.match_cap:            # a frame against itself: more matches than LF_MAX_MATCHES -- refused loudly
            with pytest.raises(capi.LinefrontError) as e:
                ctx.line_matching_node_pair(c["query"], 100 + c["ids"][0], c["train"], 100 + c["ids"][1], adjacent=c["adjacent"],
                                            cap=len(c["query"]))
            assert e.value.status == capi.LF_ERR_CAPACITY
            over += 1
            continue
        mq, mt, md = ctx.line_matching_node_pair(c["query"], 100 + c["ids"][0], c["train"], 100 + c["ids"][1], adjacent=c["adjacent"],
                                                 cap=max(len(c["query"]), 1))
        assert np.array_equal(mq, c["mq"]) and np.array_equal(mt, c["mt"]), k
        assert np.array_equal(md, c["md"]), k                   # distances bit for bit (OpenCV's four-per-trip sum)
        tot += len(mq)
    assert tot > 1800 and over == 2
    ctx.close()
