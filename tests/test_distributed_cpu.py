"""Multi-rank layout on CPU (gloo, world_size 2): contiguous image shards, ragged
shards, and the all-gather of decoded sequences.  The decode itself is replaced by the
CPU oracle here - these tests cover the sharding and the collective, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image_caption_amd.distributed import all_gather_sequences, shard_bounds


def test_shard_bounds_cover_and_partition():
    for total in (1, 2, 7, 8, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and b >= a
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import caption_oracle as co
        dims = dict(F=16, D=8, E=8, H=12, V=31)
        p = co.make_params(0, **dims)
        img = co.make_images(1, total, 4, dims["F"])
        lo, hi = shard_bounds(total, world, rank)
        local = co.predict_image_caption_sequence(p, img[lo:hi], 3, 5, 3, 4, 1.0, "repeat")
        gathered = all_gather_sequences(torch.from_numpy(local), total=total)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), gathered.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 7])
def test_two_rank_sharded_decode_equals_single_rank(tmp_path, total):
    from oracle import caption_oracle as co
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    dims = dict(F=16, D=8, E=8, H=12, V=31)
    p = co.make_params(0, **dims)
    img = co.make_images(1, total, 4, dims["F"])
    ref = co.predict_image_caption_sequence(p, img, 3, 5, 3, 4, 1.0, "repeat")
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        np.testing.assert_array_equal(got, ref)


def test_all_gather_is_identity_without_a_process_group():
    t = torch.arange(6, dtype=torch.int32).reshape(2, 3)
    assert all_gather_sequences(t) is t
