# SPDX-License-Identifier: Apache-2.0
"""N > 1: block-row shards per rank.  world_size-2 gloo processes on CPU, each rank
compressing its shard with the scalar build of the kernel source; the gathered stream must equal the
single-context whole-image stream byte for byte.  (On GPUs bench.py uses the same rank/world plumbing
with one image per rank; there is no data-path collective to test.)"""
import os
import socket
import sys
import oracle_libs as O  # (path set up by conftest.py)

import numpy as np
import pytest

import images

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_row_shard_partitions_rows_exactly(A):
    for dim_y in (1, 5, 6, 7, 36, 37, 94, 8192):
        for block_y in (4, 5, 6, 8, 10, 12):
            for world in (1, 2, 3, 4, 8):
                blocks_y = (dim_y + block_y - 1) // block_y
                rows, texels = [], []
                for r in range(world):
                    row0, row1, y0, y1 = A.block_row_shard(dim_y, block_y, r, world)
                    assert 0 <= row0 <= row1 <= blocks_y
                    assert y0 == min(row0 * block_y, dim_y) and y1 == min(row1 * block_y, dim_y)
                    rows += list(range(row0, row1))
                    texels += list(range(y0, y1))
                assert rows == list(range(blocks_y))
                assert texels == list(range(dim_y))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, block, size, quality, result_path):
    sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import astcenc_amd as A
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        lib = A.Library(O.LIB_EMU)
        img = A.synthetic_image(size[0], size[1], 11)
        err, cfg = lib.config_init(A.PRF_LDR, block[0], block[1], 1, quality, 0)
        assert err == 0
        err, ctx = lib.context_alloc(cfg, 1)
        assert err == 0
        offset, part = A.compress_shard(lib, ctx, img, block, rank, world)
        lib.context_free(ctx)
        # control-plane gather of (offset, bytes) to rank 0 -- test plumbing, not part of the data path
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((offset, part.tobytes()), gathered, dst=0)
        if rank == 0:
            blocks_x, blocks_y = (size[0] + block[0] - 1) // block[0], (size[1] + block[1] - 1) // block[1]
            out = np.zeros(blocks_x * blocks_y * 16, dtype=np.uint8)
            for off, data in gathered:
                out[off: off + len(data)] = np.frombuffer(data, dtype=np.uint8)
            np.save(result_path, out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("block,size,quality", [((6, 6), (50, 94), 60.0), ((8, 5), (41, 23), 10.0), ((12, 12), (30, 10), 10.0)])
def test_two_rank_gloo_shards_equal_whole_image(emu, A, tmp_path, block, size, quality):
    import torch.multiprocessing as mp
    result = str(tmp_path / "sharded.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, block, size, quality, result), nprocs=2, join=True)
    got = np.load(result)
    want = emu.compress(A.synthetic_image(size[0], size[1], 11), block, quality)
    assert got.tobytes() == want.tobytes()
