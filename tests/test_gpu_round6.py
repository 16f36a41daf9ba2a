"""Round 6: the fp32-accurate split mode (FFH_MATH_FP32_SPLIT_BF16X3) fed from producer-kept THREE-PLANE IMAGES
(ffh_ctx_bf16x3_mirror_set, ABI 14; csrc/linear_x3_dma.hip).  No reference counterpart for the mode (the reference's GEMMs are cuBLAS fp32
[ref: src/ops/linear.cu:436-453,624-659]; precedent for a math mode behind a handle switch: src/runtime/model.cu:81-83): the mode is held to the
SAME bound as the exact-fp32 kernels -- 1e-5 of the term mass against float64 / the fp32 oracle -- and the images to the bit:
  * the image format ("I32", include/ff_hip.h) against an independent numpy restatement and the oracle's restatement;
  * every producer's image (GEMM epilogues, the gather, the optimizers, the one-launch narrow backward, the explicit conversion);
  * the LDS-DMA GEMMs on the layers of the benched step and on ragged shapes, with the route asserted;
  * images on against images off (the split-in-kernel form) on the same operands.
"""
import numpy as np
import pytest

from dlrm_flexflow_amd import capi

MATH_X3 = 3        # FFH_MATH_FP32_SPLIT_BF16X3_ALL: every wide layer on the split kernels, whatever its size (the mode proper, 2, leaves small GEMMs to the fp32 kernels)


# ---------------------------------------------------------------------------------------------------------------------
# numpy restatement of the split and of the image (independent of csrc/ and of oracle/)
# ---------------------------------------------------------------------------------------------------------------------
def bf16_bits(a):
    """float32 -> bfloat16 bits (round to nearest even)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def bits_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def split3(a):
    a = np.ascontiguousarray(a, np.float32)
    b1 = bf16_bits(a); r1 = a - bits_f32(b1)          # exact in fp32
    b2 = bf16_bits(r1); r2 = r1 - bits_f32(b2)
    b3 = bf16_bits(r2)
    return b1, b2, b3


def image_of(flat):
    """the I32 image of a flat float32 array whose element 0 starts a group and whose length is a multiple of 32: uint16 [groups][3][32]"""
    b1, b2, b3 = split3(flat)
    return np.stack([b1.reshape(-1, 32), b2.reshape(-1, 32), b3.reshape(-1, 32)], axis=1)


def test_split3_restatement_is_fp32_sized():
    rng = np.random.default_rng(3)
    a = (rng.uniform(-1, 1, 4096) * 2.0 ** rng.integers(-30, 30, 4096)).astype(np.float32)
    b1, b2, b3 = split3(a)
    s = bits_f32(b1).astype(np.float64) + bits_f32(b2).astype(np.float64) + bits_f32(b3).astype(np.float64)
    assert np.all(np.abs(s - a.astype(np.float64)) <= 2.0 ** -24 * np.abs(a.astype(np.float64)))


def test_oracle_restates_the_image(oracle):
    """oracle/ffh_oracle.c ffh_convert_f32_to_bf16x3 == the numpy restatement (CPU; the GPU producers are compared with both below)"""
    import ctypes as C
    lib = oracle.lib()
    rng = np.random.default_rng(5)
    n = 32 * 40
    buf = np.zeros(n + 64, np.float32)
    off = (-buf.ctypes.data // 4) % 32          # first 128-byte aligned element
    a = buf[off:off + n]
    a[:] = (rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-12, 12, n)).astype(np.float32)
    raw = np.zeros(n // 32 * 96 + 64, np.uint16)
    ioff = (-raw.ctypes.data // 2) % 64
    img = raw[ioff:ioff + n // 32 * 96]
    assert lib.lib.ffh_ctx_bf16x3_mirror_set(lib.ctx, a.ctypes.data, n * 4, img.ctypes.data) == 0
    try:
        assert lib.lib.ffh_convert_f32_to_bf16x3(lib.ctx, a.ctypes.data, 1, n, n, None) == 0
        assert np.array_equal(img.reshape(-1, 3, 32), image_of(a))
        img[:] = 0
        # a 5 x 40 sub-matrix at column 7 of rows 64 apart: only its elements change
        assert lib.lib.ffh_convert_f32_to_bf16x3(lib.ctx, a[7:].ctypes.data, 5, 40, 64, None) == 0
        full = image_of(a)
        mask = np.zeros(n, bool)
        for r in range(5):
            mask[7 + 64 * r:7 + 64 * r + 40] = True
        m3 = np.broadcast_to(mask.reshape(-1, 1, 32), full.shape)
        assert np.array_equal(img.reshape(-1, 3, 32)[m3], full[m3]) and not img.reshape(-1, 3, 32)[~m3].any()
        assert lib.lib.ffh_convert_f32_to_bf16x3(lib.ctx, a.ctypes.data, 1, n + 1, n + 1, None) != 0          # reaches past the region
    finally:
        assert lib.lib.ffh_ctx_bf16x3_mirror_set(lib.ctx, a.ctypes.data, n * 4, None) == 0


# ---------------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def hip_x3(hip):
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, MATH_X3) == 0
    yield hip
    assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0) == 0


class Images:
    """registers three-plane images for a set of torch float32 tensors on one ctx; unregisters on exit"""

    def __init__(self, hip, **tensors):
        import torch
        self.hip, self.t, self.img = hip, tensors, {}
        for name, t in tensors.items():
            assert t.data_ptr() % 128 == 0 and t.is_contiguous()
            self.img[name] = torch.zeros((t.numel() + 31) // 32 * 96, dtype=torch.int16, device=t.device)
            assert hip.lib.ffh_ctx_bf16x3_mirror_set(hip.ctx, t.data_ptr(), t.numel() * 4, self.img[name].data_ptr()) == 0

    def convert(self, *names):
        for n in names:
            t = self.t[n]
            self.hip.call("ffh_convert_f32_to_bf16x3", t, 1, t.numel(), t.numel(), None)

    def host(self, name):
        return self.img[name].cpu().numpy().view(np.uint16).reshape(-1, 3, 32)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for t in self.t.values():
            assert self.hip.lib.ffh_ctx_bf16x3_mirror_set(self.hip.ctx, t.data_ptr(), t.numel() * 4, None) == 0


@pytest.mark.gpu
def test_explicit_conversion_writes_the_image_bit_for_bit(hip):
    """ffh_convert_f32_to_bf16x3 (any math mode): flat ranges at aligned and unaligned starts, a sub-matrix with a leading dimension, wide
    dynamic range, signed zeros; untouched elements stay untouched."""
    import torch
    rng = np.random.default_rng(11)
    n = 32 * 1000
    a = (rng.uniform(-1, 1, n) * 2.0 ** rng.integers(-40, 40, n)).astype(np.float32)
    a[::97] = 0.0; a[5::101] = -0.0
    t = torch.from_numpy(a).to("cuda:0")
    with Images(hip, a=t) as im:
        im.convert("a")
        assert np.array_equal(im.host("a"), image_of(a))
        im.img["a"].zero_()
        hip.call("ffh_convert_f32_to_bf16x3", t.data_ptr() + 4 * 5, 1, 1001, 1001, None)         # unaligned flat range: the scalar path
        hip.call("ffh_convert_f32_to_bf16x3", t.data_ptr() + 4 * 4096, 37, 96, 128, None)        # 37 x 96 at leading dimension 128: the vector path
        mask = np.zeros(n, bool); mask[5:1006] = True
        for r in range(37):
            mask[4096 + 128 * r:4096 + 128 * r + 96] = True
        full, got = image_of(a), im.host("a")
        m3 = np.broadcast_to(mask.reshape(-1, 1, 32), full.shape)
        assert np.array_equal(got[m3], full[m3]) and not got[~m3].any()
    # outside every registered region: refused
    assert hip.lib.ffh_convert_f32_to_bf16x3(hip.ctx, t.data_ptr(), 1, 32, 32, None) != 0


LAYERS = [(16384, 1024, 1024), (16416, 1024, 1056), (32768, 1024, 512), (16384, 1088, 1344)]


@pytest.mark.gpu
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("B,IN,OUT", LAYERS + [(32768, 3456, 1024)])
def test_x3_dma_gemms_from_images_meet_the_fp32_bound_and_write_images(hip_x3, B, IN, OUT):
    """The LDS-DMA split GEMMs (csrc/linear_x3_dma.hip) on layers of the benched step and on ragged shapes (edge tiles in rows and columns),
    images registered as the model registers them: forward with bias + relu, then the model's backward form (premasked dy, dX stored and
    masked by relu'(x), weight gradient on its own stream) against float64 sums of the fp32 operands at 1e-5 of the term mass -- the bound of
    the exact-fp32 kernels; the images of y and dX are the images of exactly the fp32 values stored; the route names the kernel for all three
    GEMMs.  Then the same calls WITHOUT images (the split-in-kernel form): same bound between the two."""
    import torch
    hip, dev = hip_x3, "cuda:0"
    rng = np.random.default_rng(B + IN + OUT)
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    y_e = np.maximum(x64 @ w64.T + b, 0)
    dy_e = np.where(y_e.astype(np.float32) > 0, gy, 0).astype(np.float32)        # premasked by the layer above
    d64 = dy_e.astype(np.float64)
    dw_e, db_e, dx_e = d64.T @ x64, d64.sum(0), np.where(x > 0, d64 @ w64, 0)
    ax, aw, ad = np.abs(x64), np.abs(w64), np.abs(d64)
    masses = {"y": ax @ aw.T + np.abs(b), "dx": ad @ aw, "dw": ad.T @ ax, "db": ad.sum(0)}
    exact = {"y": y_e, "dx": dx_e, "dw": dw_e, "db": db_e}
    xd, wd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)

    def run(images):
        y = torch.full((B, OUT), 3.0, device=dev); dx = torch.full((B, IN), 9.0, device=dev)
        dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev); dy = torch.from_numpy(dy_e).to(dev)
        im = Images(hip, x=xd, w=wd, y=y, dy=dy, dx=dx) if images else None
        try:
            if im:
                im.convert("x", "w", "dy")
            hip.call("ffh_linear_fwd", xd, IN, y, OUT, wd, bd, IN, OUT, B, capi.AC_MODE_RELU, None)
            r_f = hip.lib.ffh_linear_last_route(hip.ctx).decode()
            flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_MASK_BY_X
            s2 = torch.cuda.Stream()
            hip.call("ffh_linear_bwd_ex", xd, IN, dx, IN, y, OUT, dy, OUT, wd, dw, db, IN, OUT, B, capi.AC_MODE_RELU, flags, None, s2.cuda_stream)
            r_b = hip.lib.ffh_linear_last_route(hip.ctx).decode()
            torch.cuda.synchronize()
            out = {"y": y.cpu().numpy(), "dx": dx.cpu().numpy(), "dw": dw.cpu().numpy(), "db": db.cpu().numpy()}
            imgs = {k: im.host(k) for k in ("y", "dx")} if im else None
        finally:
            if im:
                im.__exit__()
        return out, imgs, r_f, r_b

    got, imgs, r_f, r_b = run(True)
    print("routes:", r_f, "|", r_b)
    assert "x3_dma_256x256_planes+image" in r_f and r_b.count("x3_dma_256x256_planes") == 2 and "x3_dma_256x256_planes+image" in r_b, (r_f, r_b)
    for k in exact:
        bad = np.abs(got[k].astype(np.float64) - exact[k]) > 1e-5 * masses[k] + 1e-6
        assert not bad.any(), f"{k}: {bad.sum()} of {bad.size} off, worst {np.abs(got[k] - exact[k]).max():.3e}"
    assert np.array_equal(imgs["y"], image_of(got["y"].ravel())), "image of y"
    assert np.array_equal(imgs["dx"], image_of(got["dx"].ravel())), "image of dx"
    if IN <= 1088:
        off, _, rf0, rb0 = run(False)
        assert "x3_dma" not in rf0 + rb0 and "bf16x3" in rf0, (rf0, rb0)
        for k in exact:
            assert np.all(np.abs(got[k].astype(np.float64) - off[k]) <= 1e-5 * masses[k] + 1e-6), k
        # forward and data gradient: the split-in-kernel form sums the same products in the same order (gemm_bf16x3_v2_kernel): the same BITS;
        # the weight gradient's k-slices meet by atomics in both forms
        assert got["y"].tobytes() == off["y"].tobytes(), "y: images on and off differ in bits"
        assert got["dx"].tobytes() == off["dx"].tobytes(), "dx: images on and off differ in bits"


@pytest.mark.gpu
def test_split_mode_bits_of_a_row_do_not_depend_on_the_kernel_that_computed_it(hip_x3):
    """Three kernels serve the split mode's forward, by tile count: the LDS-DMA kernel from images (16384 samples of 1024 -> 1024: 256 tiles of
    256 x 256), the split-in-kernel form on four waves of 64 x 64 (8192 samples: 512 tiles of 128 x 128) and on eight waves of 32 x 64 (4096
    samples: 256 tiles, one workgroup per CU).  All three add the same six products per k-step in the same order, and a row of y depends on
    its own row of x only: the first 4096 / 8192 rows of the big call are the small calls' results BIT FOR BIT (and so are the data gradients)."""
    import torch
    hip, dev = hip_x3, "cuda:0"
    IN = OUT = 1024
    rng = np.random.default_rng(77)
    x = np.maximum(rng.uniform(-1, 1, (16384, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
    b = rng.uniform(-1, 1, OUT).astype(np.float32)
    dy = (rng.uniform(-1, 1, (16384, OUT)) / 16384).astype(np.float32)
    wd, bd = torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    res, routes = {}, {}
    for B in (16384, 8192, 4096):
        xd = torch.from_numpy(x[:B]).to(dev); dyd = torch.from_numpy(dy[:B]).to(dev)
        y = torch.full((B, OUT), 3.0, device=dev); dx = torch.full((B, IN), 9.0, device=dev)
        with Images(hip, x=xd, w=wd, y=y, dy=dyd, dx=dx) as im:
            im.convert("x", "w", "dy")
            hip.call("ffh_linear_fwd", xd, IN, y, OUT, wd, bd, IN, OUT, B, capi.AC_MODE_RELU, None)
            r_f = hip.lib.ffh_linear_last_route(hip.ctx).decode()
            flags = capi.LINEAR_ONLY_DX | capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DY_PREMASKED | capi.LINEAR_DX_MASK_BY_X
            hip.call("ffh_linear_bwd_ex", xd, IN, dx, IN, y, OUT, dyd, OUT, wd, torch.zeros(OUT, IN, device=dev), None, IN, OUT, B, capi.AC_MODE_RELU, flags, None, None)
            r_b = hip.lib.ffh_linear_last_route(hip.ctx).decode()
            torch.cuda.synchronize()
            res[B] = (y.cpu().numpy(), dx.cpu().numpy()); routes[B] = (r_f, r_b)
    print(routes)
    assert "x3_dma_256x256" in routes[16384][0] and "x3_dma_256x256" in routes[16384][1], routes
    assert "bf16x3_128x128" in routes[8192][0] and "bf16x3_128x128" in routes[4096][0], routes
    for B in (8192, 4096):
        assert res[16384][0][:B].tobytes() == res[B][0].tobytes(), f"y: the first {B} rows differ from the {B}-sample call"
        assert res[16384][1][:B].tobytes() == res[B][1].tobytes(), f"dx: the first {B} rows differ from the {B}-sample call"


@pytest.mark.gpu
def test_x3_weight_gradient_through_slots_is_bit_reproducible(hip_x3):
    """Where the k-slices of the LDS-DMA weight gradient fit one round of workgroups they meet through the stream's reserved slots and an
    ordered pass (route token "|slots") instead of float atomics: dW of 16384 x 1024 -> 1024 and of a ragged 12320 x 1056 -> 800, four launches
    each from a dW that already holds values -- the same bits every time, and the float64 sums at the usual bound."""
    import torch
    hip, dev = hip_x3, "cuda:0"
    for B, IN, OUT in ((16384, 1024, 1024), (12320, 1056, 800)):
        rng = np.random.default_rng(B + OUT)
        x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
        dy = (rng.uniform(-1, 1, (B, OUT)) / B).astype(np.float32)
        w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
        dw0 = rng.uniform(-1, 1, (OUT, IN)).astype(np.float32)
        xd, dyd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(dy).to(dev), torch.from_numpy(w).to(dev)
        y = torch.zeros(B, OUT, device=dev)
        outs = []
        with Images(hip, x=xd, dy=dyd) as im:
            im.convert("x", "dy")
            for rep in range(4):
                dw = torch.from_numpy(dw0).to(dev)
                flags = capi.LINEAR_ONLY_DW | capi.LINEAR_DY_PREMASKED
                hip.call("ffh_linear_bwd_ex", xd, IN, None, IN, y, OUT, dyd, OUT, wd, dw, None, IN, OUT, B, capi.AC_MODE_RELU, flags, None, None)
                route = hip.lib.ffh_linear_last_route(hip.ctx).decode()
                assert "x3_dma_256x256" in route and "|slots" in route, route
                torch.cuda.synchronize()
                outs.append(dw.cpu().numpy())
        for o in outs[1:]:
            assert o.tobytes() == outs[0].tobytes(), f"{B}x{IN}->{OUT}: two launches of the slots form differ"
        exact = dw0.astype(np.float64) + dy.astype(np.float64).T @ x.astype(np.float64)
        mass = np.abs(dw0).astype(np.float64) + np.abs(dy).astype(np.float64).T @ np.abs(x).astype(np.float64)
        assert np.all(np.abs(outs[0] - exact) <= 1e-5 * mass + 1e-6)


@pytest.mark.gpu
def test_x3_dma_declines_what_it_cannot_serve_and_the_fallback_keeps_the_image(hip_x3, oracle):
    """Shapes the LDS-DMA form does not take (too few tiles, a reduction depth that is not a multiple of 32, an operand that does not start a
    group): the split-in-kernel form runs, the result meets the same bound, and the registered image of y is still the image of what was
    stored (written by a pass over y)."""
    import torch
    hip, dev = hip_x3, "cuda:0"
    for B, IN, OUT, xoff in ((512, 1024, 512, 0), (16384, 1000, 1024, 0), (16384, 1024, 1024, 8)):
        rng = np.random.default_rng(B + IN)
        x = rng.uniform(-1, 1, (B, IN)).astype(np.float32)
        w = (rng.uniform(-1, 1, (OUT, IN)) / np.sqrt(IN)).astype(np.float32)
        xbuf = torch.zeros(B * IN + 64, device=dev)
        xd = xbuf[xoff:xoff + B * IN].view(B, IN); xd.copy_(torch.from_numpy(x))
        wd = torch.from_numpy(w).to(dev)
        y = torch.full((B, OUT), 3.0, device=dev)
        with Images(hip, x=xbuf, w=wd, y=y) as im:
            im.convert("x", "w")
            hip.call("ffh_linear_fwd", xd, IN, y, OUT, wd, None, IN, OUT, B, capi.AC_MODE_NONE, None)
            route = hip.lib.ffh_linear_last_route(hip.ctx).decode()
            torch.cuda.synchronize()
            assert "x3_dma" not in route and "bf16x3" in route, route
            yh = y.cpu().numpy()
            mass = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
            assert np.all(np.abs(yh - x.astype(np.float64) @ w.astype(np.float64).T) <= 1e-5 * mass + 1e-6)
            assert np.array_equal(im.host("y"), image_of(yh.ravel()))


@pytest.mark.gpu
def test_gather_optimizers_and_the_narrow_backward_keep_their_images(hip_x3, oracle):
    """The other producers of the list in include/ff_hip.h, each into a registered region in split mode: the image is the image of exactly the
    fp32 values the call stored (which are the oracle's, bit for bit, for the gather and the optimizers)."""
    import torch
    hip, dev = hip_x3, "cuda:0"
    rng = np.random.default_rng(17)
    # the gather into a [B][384] concat buffer: three tables of width 128 at columns 0 / 128 / 256
    B, D = 2048, 128
    Z = torch.full((B, 3 * D), -5.0, device=dev)
    entries, ws_, idxs = [], [], []
    for t, R in enumerate((1000, 37, 50000)):
        wt = rng.uniform(-1, 1, (R, D)).astype(np.float32); idx = rng.integers(0, R, (B, 2))
        ws_.append(wt); idxs.append(idx)
        entries.append((torch.from_numpy(idx).to(dev), torch.from_numpy(wt).to(dev), Z[:, t * D:], R, Z.shape[1]))
    with Images(hip, z=Z) as im:
        hip.check(hip.lib.ffh_embedding_fwd_multi(hip.ctx, hip.emb_tables(entries), 3, 2, D, B, capi.AGGR_MODE_SUM, None), "fwd_multi")
        torch.cuda.synchronize()
        z = Z.cpu().numpy()
        for t in range(3):
            assert np.array_equal(z[:, t * D:(t + 1) * D].view(np.uint32), oracle.embedding_fwd(idxs[t], ws_[t], aggr=capi.AGGR_MODE_SUM).view(np.uint32))
        assert np.array_equal(im.host("z"), image_of(z.ravel()))
    # the slab optimizers on a sub-range of a registered slab
    n = 32 * 3000
    w0 = rng.uniform(-1, 1, n).astype(np.float32); g = rng.uniform(-1, 1, n).astype(np.float32)
    for kind in ("sgd", "adam"):
        wd, gd = torch.from_numpy(w0).to(dev), torch.from_numpy(g).to(dev)
        m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        with Images(hip, w=wd) as im:
            im.convert("w")
            lo, cnt = 32 * 7 + 4, 32 * 2000 + 8          # a 16-byte aligned range that starts and ends inside groups
            if kind == "sgd":
                hip.call("ffh_sgd_update_ex", wd.data_ptr() + 4 * lo, gd.data_ptr() + 4 * lo, None, cnt, 0.01, 0.0, 0.0, 0, 0, None)
            else:
                hip.call("ffh_adam_update", wd.data_ptr() + 4 * lo, gd.data_ptr() + 4 * lo, m.data_ptr() + 4 * lo, v.data_ptr() + 4 * lo, cnt, 0.001, 0.9, 0.999, 0.0, 1e-8, 0, None)
            torch.cuda.synchronize()
            wh = wd.cpu().numpy()
            assert not np.array_equal(wh[lo:lo + cnt], w0[lo:lo + cnt]) and np.array_equal(wh[:lo], w0[:lo]) and np.array_equal(wh[lo + cnt:], w0[lo + cnt:])
            assert np.array_equal(im.host("w"), image_of(wh))
    # the one-launch backward of a narrow layer (256 -> 1 on top of the Terabyte MLP): the image of its dX, the dy of the layer below
    B, IN, OUT = 4128, 256, 1
    x = np.maximum(rng.uniform(-1, 1, (B, IN)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (OUT, IN)) / 16).astype(np.float32)
    y = oracle.linear_fwd(x, w, None, capi.AC_MODE_NONE)
    gy = rng.uniform(-1, 1, (B, OUT)).astype(np.float32)
    dx = torch.full((B, IN), 9.0, device=dev); dw = torch.zeros(OUT, IN, device=dev); db = torch.zeros(OUT, device=dev)
    with Images(hip, dx=dx) as im:
        flags = capi.LINEAR_DX_OVERWRITE | capi.LINEAR_DX_MASK_BY_X
        hip.call("ffh_linear_bwd_ex", torch.from_numpy(x).to(dev), IN, dx, IN, torch.from_numpy(y).to(dev), OUT, torch.from_numpy(gy).to(dev), OUT,
                 torch.from_numpy(w).to(dev), dw, db, IN, OUT, B, capi.AC_MODE_NONE, flags, None, None)
        assert "skinny" in hip.lib.ffh_linear_last_route(hip.ctx).decode()
        torch.cuda.synchronize()
        assert np.array_equal(im.host("dx"), image_of(dx.cpu().numpy().ravel()))


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_dlrm_step_split_mode_images_on_equals_images_off_and_the_fp32_oracle(hip, oracle):
    """Whole model in split mode at the Terabyte widths (top 3456-1024-1024-512-256-1: every big layer reads images, the gather writes the image
    of the Concat output, the optimizer the weights' image), batch 16384, rows capped: three steps with the images against the same run with
    --no-bf16-twins (operands split inside the kernels) and against the oracle backend in its DEFAULT fp32 mode, at the tolerance of the fp32
    driver tests."""
    import dlrm_helpers as H  # noqa: F401
    from dlrm_flexflow_amd import ffmodel
    rows = "-".join(str(min(r, 20000)) for r in [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10, 2208, 11938, 155, 4, 976, 14,
                                                 39979771, 25641295, 39664984, 585935, 12972, 108, 36])
    base = ["-b", "16384", "--arch-sparse-feature-size", "128", "--arch-embedding-size", rows, "--arch-mlp-bot", "13-512-256-128",
            "--arch-mlp-top", "3456-1024-1024-512-256-1", "--data-size", "16384"]
    out = {}
    for name, args in (("images", ["--backend", capi.HIP_LIB_PATH, "--fp32-split-bf16x3"]), ("in_kernel", ["--backend", capi.HIP_LIB_PATH, "--fp32-split-bf16x3", "--no-bf16-twins"]),
                       ("oracle", ["--backend", oracle.ORACLE_LIB])):
        app = ffmodel.DLRM(args + base)
        app.warmup(); app.train_steps(3, trace=False); app.model.sync()
        m = app.model
        o = {f"{m.layer_name(l)}/{i}": m.parameter(l, i).get_weights() for l in range(m.num_layers) for i in range(m.layer_num_weights(l)) if "mbedding" not in m.layer_name(l)}
        o["pred"] = m.layer_output(m.num_layers - 1).get()
        out[name] = o
        app.close()
    for k in out["images"]:
        np.testing.assert_allclose(out["images"][k], out["in_kernel"][k], rtol=2e-5, atol=2e-6, err_msg=k)
        np.testing.assert_allclose(out["images"][k], out["oracle"][k], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.gpu
def test_split_mode_proper_leaves_small_gemms_to_the_fp32_kernels(hip):
    """FFH_MATH_FP32_SPLIT_BF16X3 (2) takes a Linear layer only where its GEMMs are at least FFH_BF16X3_MIN_FLOP (2 * batch * in * out >= 1e10):
    below that the exact-fp32 kernels are the faster fp32-accurate form; ..._ALL (3) takes every wide layer.  Told by the route."""
    import torch
    dev = "cuda:0"
    route = lambda: hip.lib.ffh_linear_last_route(hip.ctx).decode()
    for B, IN, OUT, want in ((4096, 512, 256, False), (4096, 1024, 1024, False), (8192, 1024, 1024, True), (4096, 3456, 1024, True)):
        x = torch.rand(B, IN, device=dev); w = torch.rand(OUT, IN, device=dev) / IN; y = torch.zeros(B, OUT, device=dev)
        try:
            assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 2) == 0
            hip.call("ffh_linear_fwd", x, IN, y, OUT, w, None, IN, OUT, B, capi.AC_MODE_NONE, None)
            assert ("bf16x3" in route() or "x3_dma" in route()) == want, (B, IN, OUT, route())
            assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 3) == 0
            hip.call("ffh_linear_fwd", x, IN, y, OUT, w, None, IN, OUT, B, capi.AC_MODE_NONE, None)
            assert "bf16x3" in route() or "x3_dma" in route(), route()
        finally:
            assert hip.lib.ffh_ctx_set_math_mode(hip.ctx, 0) == 0
        torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("widths,B", [((13, 512, 256, 128), 4096), ((13, 512, 256, 128), 8192), ((432, 512, 256), 2048), ((64, 512, 256, 64, 16), 2048), ((16, 48, 20), 77)])
def test_mlp_chain_backward_in_deterministic_mode_is_bit_reproducible(hip, oracle, widths, B):
    """ffh_mlp_chain_bwd under ffh_ctx_set_deterministic (round 6: the batch splits of a 64 x 64 weight-gradient block meet in the stream's
    scratch and a second launch adds them in split order -- no floating-point atomics [ref: the per-layer path it replaces,
    src/ops/linear.cu:610-660]): six launches on the same inputs give the same bits in every dW / db / dy / dx, the route says "ordered", and
    the values agree with the atomic form of the same kernels to 1e-5 of the term mass."""
    import torch
    dev = "cuda:0"
    RELU = capi.AC_MODE_RELU
    rng = np.random.default_rng(sum(widths) + B)
    n = len(widths) - 1
    ws = [(rng.uniform(-1, 1, (widths[l + 1], widths[l])) / np.sqrt(widths[l])).astype(np.float32) for l in range(n)]
    bs = [rng.uniform(-0.5, 0.5, widths[l + 1]).astype(np.float32) for l in range(n)]
    x = np.maximum(rng.uniform(-1, 1, (B, widths[0])), 0).astype(np.float32)
    ys, cur = [], x
    for l in range(n):
        cur = oracle.linear_fwd(cur, ws[l], bs[l], RELU)
        ys.append(cur)
    g_top = (rng.uniform(-1, 1, (B, widths[-1])) / B).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xd, wd, yd = t(x), [t(w) for w in ws], [t(y) for y in ys]

    def launch():
        dyd = [torch.full((B, widths[l + 1]), 5.0, device=dev) for l in range(n)]
        dyd[-1] = t(g_top)
        dwd = [torch.zeros(widths[l + 1], widths[l], device=dev) for l in range(n)]
        dbd = [torch.zeros(widths[l + 1], device=dev) for l in range(n)]
        dxd = torch.zeros(B, widths[0], device=dev) if widths[0] % 4 == 0 else None
        layers = hip.chain_layers([dict(w=wd[l], y=yd[l], dy=dyd[l], dw=dwd[l], db=dbd[l], in_dim=widths[l], out_dim=widths[l + 1], activation=RELU) for l in range(n)])
        flags = capi.LINEAR_DX_OVERWRITE if dxd is not None else 0
        hip.check(hip.lib.ffh_mlp_chain_bwd(hip.ctx, capi.ptr(xd), widths[0], capi.ptr(dxd), widths[0], layers, n, B, flags, None), "chain bwd")
        route = hip.lib.ffh_linear_last_route(hip.ctx).decode()
        torch.cuda.synchronize()
        out = {f"dw{l}": dwd[l].cpu().numpy() for l in range(n)}
        out.update({f"db{l}": dbd[l].cpu().numpy() for l in range(n)})
        out.update({f"dy{l}": dyd[l].cpu().numpy() for l in range(n)})
        if dxd is not None:
            out["dx"] = dxd.cpu().numpy()
        return out, route

    ref, r0 = launch()
    assert "ordered" not in r0 and "mlp_chain_dw" in r0, r0
    hip.check(hip.lib.ffh_ctx_set_deterministic(hip.ctx, 1), "deterministic")
    try:
        runs = [launch() for _ in range(6)]
    finally:
        hip.check(hip.lib.ffh_ctx_set_deterministic(hip.ctx, 0), "deterministic")
    for out, route in runs:
        assert "mlp_chain_dw" in route and "|ordered" in route and "mlp_chain_dx" in route, route
        for k in out:
            assert out[k].tobytes() == runs[0][0][k].tobytes(), f"{k}: two deterministic launches differ"
    det = runs[0][0]
    for l in range(n):
        a = np.abs(det[f"dy{l}"]).astype(np.float64)
        xin = np.abs(x if l == 0 else ys[l - 1]).astype(np.float64)
        assert np.array_equal(det[f"dy{l}"], ref[f"dy{l}"])                            # the data-gradient chain has no atomics in either mode
        assert np.all(np.abs(det[f"dw{l}"].astype(np.float64) - ref[f"dw{l}"]) <= 1e-5 * (a.T @ xin) + 1e-6), f"dw{l}"
        assert np.all(np.abs(det[f"db{l}"].astype(np.float64) - ref[f"db{l}"]) <= 1e-5 * a.sum(0) + 1e-6), f"db{l}"


@pytest.mark.gpu
def test_sum_slices_is_the_ordered_chain(hip, oracle):
    """ffh_sum_slices_f32 (the local step of the direct all-reduce): dst[i] = ((s0[i] + s1[i]) + s2[i]) + ... in slice order, bit for bit the numpy
    chain and the oracle's, vector and scalar paths, in place on slice 0."""
    import torch
    rng = np.random.default_rng(23)
    for n, stride, G in ((4096, 4096, 8), (1001, 1003, 5), (12, 16, 2), (7, 7, 1)):
        src = (rng.uniform(-1, 1, (G, stride)) * 2.0 ** rng.integers(-8, 8, (G, stride))).astype(np.float32)
        exp = src[0, :n].copy()
        for q in range(1, G):
            exp = exp + src[q, :n]
        t = torch.from_numpy(src).to("cuda:0")
        hip.call("ffh_sum_slices_f32", t, t, G, n, stride, None)
        torch.cuda.synchronize()
        got = t.cpu().numpy()
        assert got[0, :n].tobytes() == exp.tobytes(), (n, stride, G)
        assert np.array_equal(got[1:], src[1:]) and np.array_equal(got[0, n:], src[0, n:])
        o = src.copy()
        lib = oracle.lib()
        assert lib.lib.ffh_sum_slices_f32(lib.ctx, o.ctypes.data, o.ctypes.data, G, n, stride, None) == 0
        assert o[0, :n].tobytes() == exp.tobytes()
