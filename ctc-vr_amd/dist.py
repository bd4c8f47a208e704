"""Multi-GPU plumbing: utterances (streams) shard across ranks with no data-path collective;
the only collective is one broadcast of the packed weight blob from the source rank (RCCL over xGMI on the
GPU box: torch.distributed backend "nccl"; gloo on CPU in the tests)."""
import numpy as np
import torch

from . import testing as T


def pack_state_dict(sd):
    """Flat float32 vector in state_dict_spec() order (num_batches_tracked slots are zero)."""
    spec = T.state_dict_spec()
    return np.concatenate([np.asarray(sd[n], np.float32).reshape(-1) if k != "nbt" else np.zeros(1, np.float32)
                           for (n, _, k) in spec])


def unpack_state_dict(flat):
    out, o = {}, 0
    for (n, s, k) in T.state_dict_spec():
        sz = int(np.prod(s)) if s else 1
        if k != "nbt":
            out[n] = np.asarray(flat[o:o + sz], np.float32).reshape(s)
        o += sz
    assert o == len(flat)
    return out


def blob_size():
    return sum(int(np.prod(s)) if s else 1 for _, s, _ in T.state_dict_spec())


def broadcast_state_dict(sd, src=0, device="cpu"):
    """Rank `src` passes its state dict (others pass None); every rank returns the same dict.
    One collective: dist.broadcast of ~88 MB float32."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sd
    if dist.get_rank() == src:
        blob = torch.from_numpy(pack_state_dict(sd)).to(device)
    else:
        blob = torch.empty(blob_size(), dtype=torch.float32, device=device)
    dist.broadcast(blob, src=src)
    return unpack_state_dict(blob.cpu().numpy())


def shard_range(n_total, rank, world):
    """Contiguous split of n_total streams: rank r owns [lo, hi)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
