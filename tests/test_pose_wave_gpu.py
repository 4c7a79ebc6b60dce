"""k_pose_w, the refinement stage in the form that shares compute units (lineslam_amd/csrc/lf_pose_wave.h; LF_POSE_WAVE=1), must
return the bits of the default resident k_pose: the pose fixtures, the pair-size sweep and repeated launches (determinism) are
run again in a process of their own with the variant selected (the library reads the switch once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)


def test_wave_form_passes_the_pose_tests(built_lib):
    r = _run(["-m", "pytest", "tests/test_pose_golden_gpu.py", "tests/test_pair_sizes_gpu.py", "tests/test_pair_gpu.py", "-m", "gpu", "-x", "-q"],
             LF_POSE_WAVE="1")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_wave_form_is_deterministic_and_equals_the_resident_form(built_lib):
    code = r"""
import sys, numpy as np, torch
from lineslam_amd import capi, synth
F = 48
g, d, _ = synth.sequence(F, seed=3)
P = capi.default_params(launch=True)
ctx = capi.Context(640, 480, max_batch=F, params=P)
dg, dd = torch.from_numpy(g).cuda(), torch.from_numpy(d).cuda()
ctx.detect3d_batch_device(dg.data_ptr(), dd.data_ptr(), F, synth.K_TUM, np.arange(F, dtype=np.uint64))
pq, pt = np.arange(1, F, dtype=np.int32), np.arange(0, F - 1, dtype=np.int32)
out = []
for rep in range(12):
    ctx.match_pairs_device(pq, pt)
    out.append(b"".join(bytes(ctx.pair_result(i, allow_overflow=True)) for i in range(F - 1)))
assert all(o == out[0] for o in out), "repeated launches differ"
for rep in range(30):                      # one pair per launch, many launches (the shape that once faulted)
    ctx.match_pairs_device(pq[:1], pt[:1])
    assert bytes(ctx.pair_result(0, allow_overflow=True)) == out[0][:len(bytes(ctx.pair_result(0, allow_overflow=True)))]
sys.stdout.buffer.write(np.frombuffer(out[0], np.uint8).tobytes().hex().encode())
"""
    a = _run(["-c", code], LF_POSE_WAVE="1")
    b = _run(["-c", code], LF_POSE_WAVE="0")
    assert a.returncode == 0, a.stderr[-3000:]
    assert b.returncode == 0, b.stderr[-3000:]
    assert a.stdout == b.stdout and len(a.stdout) > 1000


@pytest.mark.parametrize("form", ["light", "heavy"])
def test_both_forms_of_k_match_pass_the_matching_tests(built_lib, form):
    """k_match exists in two forms with the same arithmetic (csrc/lf_pair.hip: light = 256 threads / 42 KB for the big batches of a
    pipelined step, heavy = 512 threads / 103 KB for launches that have the chip to themselves); the library picks by launch size,
    LF_MATCH_FORM forces one: the matching fixtures, the loop-closure launches and the pair tests under each."""
    r = _run(["-m", "pytest", "tests/test_match_golden_gpu.py", "tests/test_loopclosure_gpu.py", "tests/test_pair_gpu.py", "tests/test_pair_sizes_gpu.py",
              "-m", "gpu", "-x", "-q"], LF_MATCH_FORM=form)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
