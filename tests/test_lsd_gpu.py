"""GPU parity (run with -m gpu on the MI355X box): HIP LSD vs the oracle, through the C ABI.

The oracle flavour "lf" evaluates the device-side transcendentals with the same lf_math.h source as
the kernels, so EVERYTHING must agree bit for bit: scaled image, angles, gradient magnitudes, seed
order, segments (doubles) and the integer region labels.  The "ref" flavour (host libm, pinned to the
reference) must agree on the integer support.
"""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

CASES = [("tum", 22.5), ("tum", 40.0), ("chairs", 22.5), ("chairs", 40.0)]


def _ctx(img, ang, batch=1):
    from lineslam_amd import capi
    p = capi.default_params()
    p.lsd_angle_th = ang
    h, w = img.shape
    return capi.Context(w, h, max_batch=batch, params=p)


@pytest.mark.parametrize("name,ang", CASES)
def test_lsd_bit_exact_vs_oracle(built_lib, fixtures_lsd, name, ang):
    img = fixtures_lsd[name]
    ctx = _ctx(img, ang)
    segs, labels = ctx.lsd(img)
    so, lo, dbg = O.lsd_oracle(img, ang, flavour="lf", debug=True)
    assert np.array_equal(ctx.lsd_debug(0, 0), dbg["scaled"])
    assert np.array_equal(ctx.lsd_debug(0, 2)[:-1, :-1], dbg["modgrad"][:-1, :-1])
    assert np.array_equal(ctx.lsd_debug(0, 1), dbg["angles"])
    assert np.array_equal(ctx.lsd_debug(0, 3).astype(np.int64), dbg["seeds"].astype(np.int64))
    assert np.array_equal(labels.astype(np.int32), lo)
    assert segs.shape == so.shape and np.array_equal(segs, so)
    # integer support also identical to the libm flavour == the reference's golden labels
    key = "%s_a%g" % (name, ang)
    assert np.array_equal(labels, fixtures_lsd[key + "_labels"])
    ctx.close()


def test_lsd_batch_of_shifted_frames(built_lib, fixtures_lsd):
    """Frames of one batch are independent: 6 different crops/shifts in one launch."""
    import torch
    base = fixtures_lsd["tum"]
    frames = [np.roll(base, (3 * k, 5 * k), axis=(0, 1)) for k in range(6)]
    frames[3] = np.ascontiguousarray(frames[3][::-1])
    batch = np.stack(frames)
    ctx = _ctx(base, 40.0, batch=6)
    d = torch.from_numpy(batch).cuda()
    ctx.lsd_batch_device(d.data_ptr(), 6)
    for k in range(6):
        so, lo = O.lsd_oracle(frames[k], 40.0, flavour="lf")
        assert np.array_equal(ctx.lsd_labels(k).astype(np.int32), lo), k
        assert np.array_equal(ctx.lsd_segments(k), so), k
    ctx.close()


def test_lsd_flat_and_noise_images(built_lib):
    """Edge cases: a constant image (no seeds, no segments) and pure noise."""
    from lineslam_amd import capi
    rng = np.random.default_rng(7)
    for img in (np.full((480, 640), 128, np.uint8), rng.integers(0, 256, (480, 640), dtype=np.uint8)):
        ctx = capi.Context(640, 480)
        segs, labels = ctx.lsd(img)
        so, lo = O.lsd_oracle(img, 22.5, flavour="lf")
        assert segs.shape == so.shape and np.array_equal(segs, so)
        assert np.array_equal(labels.astype(np.int32), lo)
        ctx.close()


@pytest.mark.parametrize("ang", [22.5, 40.0])
def test_nfa_table_equals_direct_evaluation(built_lib, monkeypatch, ang):
    """rect_improve reads nfa(n, k, level) of small rectangles from a table filled by k_nfa_table at context
    creation; LF_NFA_TABLE=0 evaluates every nfa() in the sweep itself.  Both must give the same frames."""
    import torch
    from lineslam_amd import capi, synth
    g, _, _ = synth.sequence(6, seed=11)
    d = torch.from_numpy(g).cuda()
    p = capi.default_params()
    p.lsd_angle_th = ang
    out = []
    for tab in ("1", "0"):
        monkeypatch.setenv("LF_NFA_TABLE", tab)
        ctx = capi.Context(640, 480, max_batch=6, params=p)
        ctx.lsd_batch_device(d.data_ptr(), 6)
        out.append([(ctx.lsd_segments(k), ctx.lsd_labels(k)) for k in range(6)])
        ctx.close()
    for (s1, l1), (s0, l0) in zip(*out):
        assert len(s1) > 50 and np.array_equal(s1, s0) and np.array_equal(l1, l0)


@pytest.mark.parametrize("w,h", [(64, 48), (33, 17), (130, 9), (16, 16)])
def test_lsd_tiny_images(built_lib, w, h):
    """Images smaller than the kernels' tiles and windows (a few scaled rows or columns): still the oracle's output."""
    from lineslam_amd import capi
    rng = np.random.default_rng(w * 100 + h)
    img = np.zeros((h, w), np.uint8)
    img[:, w // 2:] = 200                                   # one vertical edge
    img[h // 3:h // 3 + max(2, h // 4), : w // 3] = 120     # and a small block
    img = np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    ctx = capi.Context(w, h, max_batch=1)
    segs, labels = ctx.lsd(img)
    so, lo = O.lsd_oracle(img, 22.5, 0.7, flavour="lf")
    assert np.array_equal(segs, so)
    assert np.array_equal(labels.astype(np.int32), lo)
    ctx.close()


def test_lsd_large_image(built_lib):
    """A 1280x720 frame (four times the bench resolution) through the same kernels."""
    from lineslam_amd import capi, synth
    g, _, _ = synth.sequence(1, seed=9, w=1280, h=720)
    ctx = capi.Context(1280, 720, max_batch=1)
    segs, labels = ctx.lsd(g[0])
    so, lo = O.lsd_oracle(g[0], 22.5, 0.7, flavour="lf")
    assert len(so) > 50
    assert np.array_equal(segs, so)
    assert np.array_equal(labels.astype(np.int32), lo)
    ctx.close()


def _one_wave_cases(fixtures_lsd):
    rng = np.random.default_rng(41)
    cases = [(fixtures_lsd[name], ang) for name, ang in CASES]
    cases.append((np.full((480, 640), 128, np.uint8), 22.5))                            # no seeds at all
    cases.append((rng.integers(0, 256, (480, 640), dtype=np.uint8), 40.0))               # noise: every seed a tiny region
    for w, h in ((64, 48), (33, 17), (130, 9), (16, 16), (100, 75)):                     # scaled widths 51, 26, 104, 12, 80: rows of the
        img = np.zeros((h, w), np.uint8)                                                 # bitmap are padded to whole words
        img[:, w // 2:] = 200
        img[h // 3:h // 3 + max(2, h // 4), : w // 3] = 120
        cases.append((np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8), 22.5))
    return cases


SWEEP_MODES = {"plain": "0", "lu": "1"}      # LF_SWEEP_LU


@pytest.mark.parametrize("mode", list(SWEEP_MODES))
def test_one_wavefront_sweeps_vs_oracle(built_lib, fixtures_lsd, monkeypatch, mode):
    """The kernels behind large batches, forced on small ones (LF_SWEEP_WAVES=1): k_lsd_sweep (global `used`; the default) and,
    with LF_SWEEP_LU=1, k_lsd_sweep_lu -- `used` + NOTDEF as a bitmap in LDS, the seeds' (cos, sin) tiles staged in LDS by DMA one
    region ahead -- for scaled images of up to 512 x 384 (k_lsd_sweep above that).  Segments and region labels bit-equal to the oracle, on the reference's own test
    images, on flat / noise frames, on images smaller than a tile, and on a 1280 x 720 frame."""
    from lineslam_amd import capi, synth
    monkeypatch.setenv("LF_SWEEP_WAVES", "1")
    monkeypatch.setenv("LF_SWEEP_LU", SWEEP_MODES[mode])
    cases = _one_wave_cases(fixtures_lsd)
    big, _, _ = synth.sequence(1, seed=9, w=1280, h=720)
    cases.append((big[0], 22.5))                                                         # 1024 x 576 scaled: not the LDS variant
    for img, ang in cases:
        p = capi.default_params(); p.lsd_angle_th = ang
        h, w = img.shape
        ctx = capi.Context(w, h, max_batch=1, params=p)
        segs, labels = ctx.lsd(img)
        so, lo = O.lsd_oracle(img, ang, flavour="lf")
        assert segs.shape == so.shape and np.array_equal(segs, so), (img.shape, ang)
        assert np.array_equal(labels.astype(np.int32), lo), (img.shape, ang)
        ctx.close()


@pytest.mark.parametrize("mode", list(SWEEP_MODES))
def test_one_wavefront_sweep_on_a_batch(built_lib, monkeypatch, mode):
    """k_lsd_sweep / k_lsd_sweep_lu on 24 different frames of one launch (LU: every frame its own LDS bitmap and tile slots), launch-file angle"""
    import torch
    from lineslam_amd import capi, synth
    monkeypatch.setenv("LF_SWEEP_WAVES", "1")
    monkeypatch.setenv("LF_SWEEP_LU", SWEEP_MODES[mode])
    g, _, _ = synth.sequence(24, seed=21)
    P = capi.default_params(launch=True)
    ctx = capi.Context(640, 480, max_batch=24, params=P)
    d = torch.from_numpy(g).cuda()
    ctx.lsd_batch_device(d.data_ptr(), 24)
    for k in range(24):
        so, lo = O.lsd_oracle(g[k], P.lsd_angle_th, P.lsd_density_th, flavour="lf")
        assert np.array_equal(ctx.lsd_segments(k), so), k
        assert np.array_equal(ctx.lsd_labels(k).astype(np.int32), lo), k
    ctx.close()
