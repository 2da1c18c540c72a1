"""tc_route_batch (csrc/route_kernels.hpp) on the GPU: the device partition of a global batch equals the host
mirror (tc_route_host) -- same owners, same shard-local slots, request order kept inside every destination --
for one destination (`only`) and for all of them, at tile / wave edge sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 300_000])
def test_device_partition_equals_host_mirror(world, n):
    import torch
    import throttlecrab_amd as t
    from throttlecrab_amd import sharded
    cap = 20_000
    rng = np.random.default_rng(world * 1000 + n)
    ids = rng.integers(0, world * cap, n).astype(np.uint32)
    if n > 10:
        ids[n // 2:n // 2 + 5] = ids[0]  # duplicates of one key: their order must survive
    owner, slot = sharded.route(ids, world, cap)
    eng = t.Engine(cap, 1 << 16)
    eng.use_torch_stream()
    d = torch.from_numpy(ids.astype(np.int32)).cuda()
    # every destination, one segment after the other
    slots, pos, counts = eng.route_batch(d, world, only=-1, want_pos=True)
    torch.cuda.synchronize()
    counts = counts.cpu().numpy()
    assert np.array_equal(counts, np.bincount(owner, minlength=world))
    slots, pos = slots.cpu().numpy().astype(np.uint32), pos.cpu().numpy()
    at = 0
    for dest in range(world):
        want = np.nonzero(owner == dest)[0]
        assert np.array_equal(pos[at:at + len(want)], want), dest          # stable: request order
        assert np.array_equal(slots[at:at + len(want)], slot[want]), dest
        at += len(want)
    # one destination only
    for dest in sorted({0, world - 1, world // 2}):
        s2, p2, c2 = eng.route_batch(d, world, only=dest, want_pos=True)
        torch.cuda.synchronize()
        want = np.nonzero(owner == dest)[0]
        assert np.array_equal(c2.cpu().numpy(), counts)
        assert np.array_equal(p2.cpu().numpy()[: len(want)], want)
        assert np.array_equal(s2.cpu().numpy().astype(np.uint32)[: len(want)], slot[want])
    eng.close()
