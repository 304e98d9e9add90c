"""-m gpu: the pinned staging buffer of t8gpu_amd/hostmem.py (opt-in upload path of the adaptive loop)."""
import numpy as np
import pytest
import torch

from t8gpu_amd import hostmem

pytestmark = pytest.mark.gpu


def test_pinned_uploader_moves_every_byte():
    """Arrays below the staging threshold, of one chunk, of several chunks and larger than the whole buffer (the uploader
    drains the stream and starts over), of several dtypes and shapes, views of C memory included."""
    up = hostmem.PinnedUploader(megabytes=8, chunk_megabytes=2)
    rng = np.random.default_rng(3)
    cases = [rng.integers(0, 1 << 30, 1000, dtype=np.int32), rng.random(300_000), rng.random((5, 700_001)).astype(np.float32),
             rng.integers(-30000, 30000, 9_000_003, dtype=np.int16), rng.random(2_500_000), np.empty(0, np.float64),
             rng.integers(0, 255, (3, 4_000_001), dtype=np.uint8)]
    outs = [up.upload(a) for a in cases]
    torch.cuda.synchronize()
    for a, t in zip(cases, outs):
        assert tuple(t.shape) == a.shape and np.array_equal(t.cpu().numpy(), a)
    # back to back with kernels on the same stream: the copies are ordered with them
    a = rng.random(3_000_000)
    t = up.upload(a)
    s = (t * 2.0).sum().item()
    assert abs(s - 2.0 * a.sum()) < 1e-6 * a.size


def test_pinned_uploader_chunk_never_exceeds_the_buffer_and_wraps_wait_for_every_stream():
    """A staging buffer smaller than the default chunk (the chunk was not clipped and the staging copy wrote
    past the pinned buffer), uploads queued on two streams in turn: a wrap waits for the copies of both."""
    up = hostmem.PinnedUploader(megabytes=1, chunk_megabytes=32)
    assert up.chunk <= up.cap
    with pytest.raises(ValueError):
        hostmem.PinnedUploader(megabytes=0)
    rng = np.random.default_rng(4)
    cases = [rng.random(700_000), rng.integers(0, 1 << 30, 1_300_001, dtype=np.int32), rng.random(400_000).astype(np.float32)]
    side = torch.cuda.Stream()
    outs = []
    for i, a in enumerate(cases * 3):
        if i % 2:
            with torch.cuda.stream(side):
                outs.append((a, up.upload(a)))
        else:
            outs.append((a, up.upload(a)))
    torch.cuda.synchronize()
    for a, t in outs:
        assert np.array_equal(t.cpu().numpy(), a)


def test_solver_through_the_pinned_uploader_matches_the_default_path():
    from _gpu import perturbed_state
    from t8gpu_amd.solver import PlainSolver
    from t8gpu_amd.synth import SynthMesh
    part = SynthMesh(3, 4, 5, band=0.1).partition()
    st = perturbed_state(part, 9)
    a = PlainSolver(part, torch.float64, mode="fused", state=st)
    saved = hostmem._uploader
    try:
        hostmem._uploader = hostmem.PinnedUploader(megabytes=4, chunk_megabytes=1)
        b = PlainSolver(part, torch.float64, mode="fused", state=st)
    finally:
        hostmem._uploader = saved
    for _ in range(3):
        a.iterate(1e-3)
        b.iterate(1e-3)
    torch.cuda.synchronize()
    assert torch.equal(a.state(), b.state()) and torch.equal(a.speed, b.speed)
