"""Device-side generators (wn_generate_*) against wavenet_amd/generators.py: bit-identical batches."""
import numpy as np
import pytest

from gpu_util import dev_empty, from_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    from wavenet_amd import Context
    c = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    torch.cuda.synchronize()
    c.close()


@pytest.mark.parametrize("R,C", [(64, 64), (16, 4), (1024, 1), (1, 32), (128, 256)])
def test_uniform_and_needle(ctx, R, C):
    import torch
    from wavenet_amd import generators as G
    B, first, seed = 7, 3, 1234
    d = dev_empty(B * R * C)
    ctx.generate_uniform(d, B, R, C, seed=seed, first=first)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev(d, B, R, C), G.uniform(first + B, R, C, seed)[first:])
    # a threshold that leaves some images of the small shapes empty exercises the kept pixel
    thr = 0.995 if R * C >= 1024 else 0.97
    ctx.generate_needle(d, B, R, C, seed=seed, first=first, threshold=thr)
    torch.cuda.synchronize()
    ref = G.needle(first + B, R, C, seed, threshold=thr)[first:]
    got = from_dev(d, B, R, C)
    if thr != 0.995:
        # the kept pixel's value is defined with the default threshold's constants; only its position and
        # the thresholded pixels are comparable here
        assert np.array_equal(got != 0, ref != 0)
    else:
        assert np.array_equal(got, ref)


def test_needle_empty_images_keep_one_pixel(ctx):
    import torch
    from wavenet_amd import generators as G
    B, R, C = 40, 8, 8    # 64 pixels at threshold 0.995: most images would be empty
    d = dev_empty(B * R * C)
    ctx.generate_needle(d, B, R, C, seed=9)
    torch.cuda.synchronize()
    got = from_dev(d, B, R, C)
    assert np.array_equal(got, G.needle(B, R, C, 9))
    assert ((got != 0).reshape(B, -1).sum(axis=1) >= 1).all()


@pytest.mark.parametrize("R,C,B", [(64, 64, 33), (512, 512, 2), (32, 128, 5)])
def test_calorimeter(ctx, R, C, B):
    import torch
    from wavenet_amd import generators as G
    first, seed = 2, 1000
    d = dev_empty(B * R * C)
    ctx.generate_calorimeter(d, B, R, C, seed=seed, first=first)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev(d, B, R, C), G.calorimeter(first + B, R, C, seed=seed)[first:])


def test_generated_batch_trains_like_an_uploaded_one(ctx):
    import torch
    from gpu_util import to_dev
    from wavenet_amd import generators as G
    import filters as F
    B, R, C = 64, 64, 64
    d = dev_empty(B * R * C)
    ctx.generate_calorimeter(d, B, R, C, seed=5)
    ctx.set_filter(F.DB4)
    ctx.train_batch(d, B, R, C)
    s1 = ctx.sums().copy()
    ctx.train_batch(to_dev(G.calorimeter(B, R, C, seed=5)), B, R, C)
    assert np.array_equal(ctx.sums(), s1)


def test_calorimeter_tables_survive_a_cost_grid():
    """One context: calorimeter batch, a cost grid large enough to (re)allocate the filter table, the
    same batch again, then a larger grid (the table grows) and the batch a third time -- bit-identical
    every time, and the context closes cleanly."""
    import torch
    from wavenet_amd import Context, generators as G
    import filters as F
    c = Context(0, stream=torch.cuda.current_stream().cuda_stream)
    B, R, C = 9, 64, 64
    ref = G.calorimeter(B, R, C, seed=77)
    d = dev_empty(B * R * C)
    S = dev_empty(600 * B)
    rng = np.random.default_rng(3)
    for nfilters in (40, 600):
        c.generate_calorimeter(d, B, R, C, seed=77)
        torch.cuda.synchronize()
        assert np.array_equal(from_dev(d, B, R, C), ref)
        filters = np.asarray(F.DB4)[None, :] * (1 + 1e-2 * rng.standard_normal((nfilters, 4)))
        c.sparse_cost_grid(d, B, R, C, filters, S)
        torch.cuda.synchronize()
        assert np.isfinite(S[:nfilters * B].cpu().numpy()).all()
    c.generate_calorimeter(d, B, R, C, seed=77)
    torch.cuda.synchronize()
    assert np.array_equal(from_dev(d, B, R, C), ref)
    c.close()


@pytest.mark.parametrize("R,C,B", [(64, 64, 7), (1024, 1, 3), (16, 256, 4)])
def test_gaussian(ctx, R, C, B):
    """The Gaussian bumps written on the device are those of generators.gaussian: the same table of exp and the same
    additions and multiplications on both sides."""
    import torch
    from wavenet_amd import generators as G
    first, seed = 3, 21
    d = dev_empty(B * R * C)
    ctx.generate_gaussian(d, B, R, C, seed=seed, first=first)
    torch.cuda.synchronize()
    ref = G.gaussian(first + B, R, C, seed=seed)[first:]
    assert np.array_equal(from_dev(d, B, R, C), ref)
    assert ref.max() <= 3.0 + 1e-12 and ref.max() > 0.5


@pytest.mark.parametrize("kind,R,C", [("gaussian", 32, 64), ("calorimeter", 64, 64), ("calorimeter", 128, 64), ("needle", 16, 16),
                                      ("uniform", 8, 32)])
def test_host_layer_generators_agree_with_their_device_forms(kind, R, C):
    """next() and nextOnDevice() of the C++ host layer's generators (tests/cpp/host_driver generate): bit-identical,
    calorimeter images of up to 4096 bins through the LDS kernel and larger ones through the one that works in
    memory.  The calorimeter images also match generators.calorimeter wherever the C library's log and exp agree
    with numpy's to the bit; a difference of an ulp in a table entry is accepted."""
    from test_host_cpp import run_driver
    from wavenet_amd import generators as G
    count, seed = 5, 1000
    out, _ = run_driver(["generate", kind, R, C, count, seed], [np.zeros(1)])
    assert np.array_equal(out["host"], out["device"])
    X = out["host"].reshape(count, C, R).transpose(0, 2, 1)        # column-major examples -> (B, R, C)
    if kind == "calorimeter":
        ref = G.calorimeter(count, R, C, seed=seed)
        assert np.allclose(X, ref, rtol=3e-7, atol=0)               # (bin contents are floats: one float ulp)
        assert (X != 0).mean() < 0.2
    elif kind == "gaussian":
        assert np.allclose(X, G.gaussian(count, R, C, seed=seed), rtol=1e-14, atol=1e-300)
    elif kind == "uniform":
        assert np.array_equal(X, G.uniform(count, R, C, seed=seed))
    else:
        assert np.array_equal(X, G.needle(count, R, C, seed=seed))
