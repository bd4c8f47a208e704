"""N>1 path on CPU: world_size-2 gloo processes exercise the weight broadcast and the stream sharding that
bench.py / the GPU ranks use (the data path itself has no collective)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ctc_vr_amd.dist as D
import ctc_vr_amd.testing as T


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = T.make_state_dict(3) if rank == 0 else None
    got = D.broadcast_state_dict(sd, src=0, device="cpu")
    ref = T.make_state_dict(3)
    same = all(np.array_equal(got[k], ref[k]) for k in got) and len(got) == 504 - 12
    # a vocabulary the receiving ranks do not know in advance: the layout travels in the header
    sd2 = T.make_state_dict(4, vocab=77, blank=3) if rank == 0 else None
    got2 = D.broadcast_state_dict(sd2, src=0, device="cpu")
    ref2 = T.make_state_dict(4, vocab=77, blank=3)
    same = same and got2["joint.ffn_out.weight"].shape == (77, 256) and all(np.array_equal(got2[k], ref2[k]) for k in got2)
    # the packed form every GPU rank hands to its context in one call (RnntEngine.load_packed): blob + vocabulary
    blob, vocab = D.broadcast_packed(T.make_state_dict(3) if rank == 0 else None, src=0, device="cpu")
    same = same and vocab == T.VOCAB and blob.dtype == torch.float32 and blob.numel() == D.blob_size(vocab) \
        and np.array_equal(blob.numpy(), D.pack_state_dict(ref))
    lo, hi = D.shard_range(130, rank, world)
    # every rank decodes only its own streams; a tiny all_gather of counts stands in for result collection
    mine = torch.tensor([hi - lo], dtype=torch.int64)
    allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, mine)
    q.put((rank, same, lo, hi, [int(c) for c in allc]))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1], "broadcast state dict differs from the source"
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 65, 65, 130)
    assert res[0][4] == [65, 65]


def test_pack_unpack_roundtrip_and_ranges():
    sd = T.make_state_dict(0)
    flat = D.pack_state_dict(sd)
    assert flat.dtype == np.float32 and flat.size == D.blob_size()
    back = D.unpack_state_dict(flat)
    assert all(np.array_equal(back[k], sd[k]) for k in back)
    sd77 = T.make_state_dict(1, vocab=77, blank=3)
    flat77 = D.pack_state_dict(sd77)
    assert flat77.size == D.blob_size(77) != D.blob_size()
    assert all(np.array_equal(v, sd77[k]) for k, v in D.unpack_state_dict(flat77, 77).items())
    import pytest
    with pytest.raises(ValueError):
        D.unpack_state_dict(flat77)                      # wrong vocabulary for this blob
    with pytest.raises(KeyError):
        D.pack_state_dict({k: v for k, v in sd.items() if k != "joint.enc_ffn.bias"})
    with pytest.raises(KeyError):
        D.pack_state_dict(dict(sd, bogus=np.zeros(3, np.float32)))
    cover = []
    for r in range(8):
        lo, hi = D.shard_range(512, r, 8)
        assert hi - lo == 64
        cover += list(range(lo, hi))
    assert cover == list(range(512))
    assert [D.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
