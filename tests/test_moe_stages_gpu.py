"""Bit-exact checks of the integer / byte stages through the C-ABI: weight packing and moe_align."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import sgl_kernel
    assert torch.cuda.is_available()
    return sgl_kernel._lib.lib()


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def packed_index_fp8(rows, cols):
    """Source element index for every packed byte (DESIGN.md §Packed weight layout), fp8 16x64 tiles."""
    ct = cols // 64
    idx = np.empty(rows * cols, dtype=np.int64)
    c = np.arange(rows * cols // 16)
    lane, tile = c & 63, c >> 6
    rt, cti = tile // ct, tile % ct
    r, g = lane & 15, lane >> 4
    base = (rt * 16 + r) * cols + cti * 64 + 8 * g
    for j in range(8):
        idx[c * 16 + j] = base + j
        idx[c * 16 + 8 + j] = base + 32 + j
    return idx


@pytest.mark.parametrize("shape", [(1, 16, 64), (3, 256, 128), (8, 1536, 2048)])
def test_pack_fp8_layout_and_roundtrip(L, shape):
    import sgl_kernel
    b, rows, cols = shape
    w = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn)
    packed = torch.ops.sgl_kernel.convert_weight_packed(w)
    assert packed.shape == w.shape and packed.dtype == w.dtype
    idx = torch.from_numpy(packed_index_fp8(rows, cols)).cuda()
    expect = w.view(torch.uint8).reshape(b, -1)[:, idx].reshape(shape)
    assert torch.equal(packed.view(torch.uint8), expect)
    back = torch.empty_like(w)
    rc = L.sglk_unpack_weight(_p(packed), _p(back), b, rows, cols, sgl_kernel._lib.W_FP8_E4M3, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(back.view(torch.uint8), w.view(torch.uint8))


@pytest.mark.parametrize("dtype,wt", [(torch.bfloat16, 0), (torch.int8, 2)])
def test_pack_roundtrip_other_dtypes(L, dtype, wt):
    w = torch.randint(-100, 100, (2, 64, 128), device="cuda").to(dtype)
    packed = torch.ops.sgl_kernel.convert_weight_packed(w)
    back = torch.empty_like(w)
    assert L.sglk_unpack_weight(_p(packed), _p(back), 2, 64, 128, wt, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(back, w) and not torch.equal(packed, w)


def test_pack_rejects_bad_shapes(L):
    w = torch.zeros(1, 16, 64, dtype=torch.uint8, device="cuda")
    o = torch.zeros_like(w)
    assert L.sglk_pack_weight(_p(w), _p(o), 1, 10, 64, 1, _stream()) == -2
    assert b"multiple" in L.sglk_last_error()
    assert L.sglk_pack_weight(_p(w), _p(w), 1, 16, 64, 1, _stream()) == -1


def align_oracle(ids, E, tile_m):
    """numpy restatement: stable counting sort by expert, ids outside [0,E) dropped."""
    flat = ids.reshape(-1)
    valid = (flat >= 0) & (flat < E)
    slots = np.nonzero(valid)[0]
    order = slots[np.argsort(flat[slots], kind="stable")]
    counts = np.bincount(flat[slots], minlength=E)
    off = np.concatenate([[0], np.cumsum(counts)])
    tiles = []
    for e in range(E):
        for i in range(0, counts[e], tile_m):
            tiles.append((e, off[e] + i, min(tile_m, counts[e] - i), 0))
    return order.astype(np.int32), off.astype(np.int32), np.array(tiles, dtype=np.int32).reshape(-1, 4)


@pytest.mark.parametrize("M,E,topk,masked", [(1, 8, 2, False), (14, 8, 8, True), (777, 128, 8, False),
                                             (4096, 128, 8, True), (3, 1024, 2, False), (20000, 16, 4, False),
                                             (1024, 128, 8, True), (1025, 256, 8, False), (130, 200, 7, True), (64, 128, 8, False)])
def test_moe_align_bit_exact(L, M, E, topk, masked):
    g = torch.Generator().manual_seed(M * 131 + E)
    ids = torch.randint(0, E, (M, topk), generator=g, dtype=torch.int32)
    if masked:
        ids[torch.rand(M, topk, generator=g) < 0.4] = -1
        ids[0, 0] = E + 5            # out-of-range ids are dropped like -1
    tile_m = 128
    S = M * topk
    d_ids = ids.cuda()
    max_tiles = L.sglk_moe_max_tiles(M, E, topk, tile_m)
    sorted_slot = torch.full((S,), -7, dtype=torch.int32, device="cuda")
    expert_off = torch.empty(E + 1, dtype=torch.int32, device="cuda")
    tile_info = torch.full((max(max_tiles, 1), 4), -7, dtype=torch.int32, device="cuda")
    num_tiles = torch.empty(1, dtype=torch.int32, device="cuda")
    ws_bytes = L.sglk_moe_align_workspace_bytes(M, E, topk)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    rc = L.sglk_moe_align(_p(d_ids), M, E, topk, tile_m, _p(sorted_slot), _p(expert_off), _p(tile_info),
                          _p(num_tiles), _p(ws), ws_bytes, _stream())
    assert rc == 0, L.sglk_last_error()
    torch.cuda.synchronize()
    order, off, tiles = align_oracle(ids.numpy(), E, tile_m)
    assert np.array_equal(expert_off.cpu().numpy(), off)
    n = int(num_tiles.item())
    assert n == len(tiles) and n <= max_tiles
    assert np.array_equal(sorted_slot.cpu().numpy()[:len(order)], order)
    assert np.array_equal(tile_info.cpu().numpy()[:n], tiles)
