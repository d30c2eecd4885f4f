"""tracyhip_pack_ragged_multi (tracy_amd/csrc/pack.hip): the used parts of fixed-stride regions back to back, kind-major -- what a rank
does to its variable-length results before the final gather.  Every alignment of source and destination, empty regions, lengths that
exceed their stride (clamped), length words read with a stride out of a record array, int32 elements; against numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import tracy_amd
    c = tracy_amd.Context(0)
    yield c
    c.close()


def expect(kinds_np, n):
    parts = []
    for buf, stride, lens, lens_stride in kinds_np:
        elem = buf.dtype.itemsize
        raw = buf.view(np.uint8)
        for i in range(n):
            ln = min(int(lens[i * lens_stride]) * elem, stride * elem)
            parts.append(raw[i * stride * elem:i * stride * elem + ln])
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pack_ragged_multi_against_numpy(ctx, seed):
    import torch
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 700))
    F = 5  # the lengths of the first two kinds are columns of an int32 record array
    rec = rng.integers(0, 50, size=(n, F)).astype(np.int32)
    s0, s1, s2 = 37, 64, 9
    rec[:, 1] = rng.integers(0, s0 + 6, size=n)   # some beyond the stride: clamped
    rec[:, 3] = rng.integers(0, s1 + 1, size=n)
    rec[rng.integers(0, n, size=max(1, n // 5)), 1] = 0  # empty regions
    lens2 = rng.integers(0, s2 + 1, size=n).astype(np.int32)
    b0 = rng.integers(0, 256, size=n * s0).astype(np.uint8)
    b1 = rng.integers(0, 256, size=n * s1).astype(np.uint8)
    b2 = rng.integers(-2**31, 2**31 - 1, size=n * s2).astype(np.int32)
    want = expect([(b0, s0, rec.reshape(-1)[1:], F), (b1, s1, rec.reshape(-1)[3:], F), (b2, s2, lens2, 1)], n)
    d_rec = torch.from_numpy(rec).cuda()
    flat = d_rec.reshape(-1)
    d = [torch.from_numpy(x).cuda() for x in (b0, b1, b2)]
    d_l2 = torch.from_numpy(lens2).cuda()
    packed, sizes = ctx.pack_ragged_multi([(d[0], s0, flat[1:], F), (d[1], s1, flat[3:], F), (d[2], s2, d_l2, 1)], n)
    assert sum(sizes) == want.size and np.array_equal(packed.cpu().numpy(), want)
    # one kind through tracyhip_pack_ragged, into a buffer that starts at an odd address
    out = torch.empty(n * s0 + 8, dtype=torch.uint8, device="cuda")
    p1, nb = ctx.pack_ragged(d[0], s0, flat[1:], n=n, lens_stride=F, out=out[3:])
    assert nb == sizes[0] and np.array_equal(p1.cpu().numpy(), want[:nb])


def test_pack_sizes_only_and_capacity_error(ctx):
    import ctypes as C
    import torch
    from tracy_amd import capi
    n, s = 40, 16
    lens = torch.arange(n, dtype=torch.int32, device="cuda") % (s + 1)
    buf = torch.arange(n * s, dtype=torch.int64, device="cuda").to(torch.uint8)
    total = int((torch.arange(n) % (s + 1)).sum())
    tot = C.c_uint64(0)
    lib = capi.lib()
    rc = lib.tracyhip_pack_ragged(ctx._h, C.c_void_p(buf.data_ptr()), C.c_uint64(s), C.c_uint32(1), C.c_void_p(lens.data_ptr()), C.c_uint32(1), C.c_uint32(n),
                                  None, C.c_uint64(0), C.byref(tot))
    assert rc == 0 and tot.value == total  # dst == NULL: the size only
    small = torch.empty(total - 1, dtype=torch.uint8, device="cuda")
    rc = lib.tracyhip_pack_ragged(ctx._h, C.c_void_p(buf.data_ptr()), C.c_uint64(s), C.c_uint32(1), C.c_void_p(lens.data_ptr()), C.c_uint32(1), C.c_uint32(n),
                                  C.c_void_p(small.data_ptr()), C.c_uint64(small.numel()), C.byref(tot))
    assert rc == capi.ERR_ARG
    assert tot.value == total
