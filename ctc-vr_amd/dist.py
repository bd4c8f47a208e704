"""Multi-GPU plumbing: utterances (streams) shard across ranks with no data-path collective;
the only collectives are a 2-word header and one broadcast of the packed weight blob from the source rank (RCCL over xGMI on
the GPU box: torch.distributed backend "nccl"; gloo on CPU in the tests)."""
import numpy as np
import torch

from . import layout as T


def vocab_of(sd):
    """Vocabulary size of a state dict in the reference's 504-key layout (rows of joint.ffn_out.weight)."""
    return int(np.asarray(sd["joint.ffn_out.weight"]).shape[0])


def pack_state_dict(sd, vocab=None):
    """Flat float32 vector in state_dict_spec(vocab) order (num_batches_tracked slots are zero).  Raises on a missing key,
    a key the layout does not know or a wrong shape: a silently different layout would corrupt the broadcast."""
    vocab = vocab_of(sd) if vocab is None else vocab
    spec = T.state_dict_spec(vocab)
    names = {n for n, _, _ in spec}
    extra = [k for k in sd if k not in names]
    if extra:
        raise KeyError(f"state dict has keys outside the 504-key layout: {extra[:4]}")
    parts = []
    for (n, shp, k) in spec:
        if k == "nbt":
            parts.append(np.zeros(1, np.float32))
            continue
        if n not in sd:
            raise KeyError(f"state dict lacks {n}")
        a = np.asarray(sd[n], np.float32)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"{n}: shape {tuple(a.shape)}, layout says {tuple(shp)}")
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


def unpack_state_dict(flat, vocab=T.VOCAB):
    out, o = {}, 0
    for (n, s, k) in T.state_dict_spec(vocab):
        sz = int(np.prod(s)) if s else 1
        if k != "nbt":
            out[n] = np.asarray(flat[o:o + sz], np.float32).reshape(s)
        o += sz
    if o != len(flat):
        raise ValueError(f"blob of {len(flat)} floats does not match the layout for vocab {vocab} ({o} floats)")
    return out


def blob_size(vocab=T.VOCAB):
    return sum(int(np.prod(s)) if s else 1 for _, s, _ in T.state_dict_spec(vocab))


def broadcast_packed(sd, src=0, device="cpu"):
    """Rank `src` passes its state dict (others pass None); every rank returns (blob, vocab): the packed float32 blob as a torch
    tensor ON `device` (it never visits the host on the receiving ranks) and the vocabulary size.  The layout is derived on the
    source rank: a 2-word header (vocabulary size, blob length) goes first, then ONE dist.broadcast of the ~88 MB blob (RCCL over
    xGMI with the "nccl" backend).  Hand the pair to RnntEngine.load_packed / StreamingBatch(packed=...): one C-ABI call."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        vocab = vocab_of(sd)
        return torch.from_numpy(pack_state_dict(sd, vocab)).to(device), vocab
    is_src = dist.get_rank() == src
    hdr_dev = device if dist.get_backend() == "nccl" else "cpu"
    if is_src:
        vocab = vocab_of(sd)
        flat = pack_state_dict(sd, vocab)
        hdr = torch.tensor([vocab, flat.size], dtype=torch.int64, device=hdr_dev)
    else:
        hdr = torch.zeros(2, dtype=torch.int64, device=hdr_dev)
    dist.broadcast(hdr, src=src)
    vocab, n = int(hdr[0]), int(hdr[1])
    if n != blob_size(vocab):
        raise ValueError(f"source rank announced {n} floats for vocab {vocab}; this rank's layout has {blob_size(vocab)}")
    bdev = device if dist.get_backend() == "nccl" else "cpu"           # gloo broadcasts host tensors
    blob = torch.from_numpy(flat).to(bdev) if is_src else torch.empty(n, dtype=torch.float32, device=bdev)
    dist.broadcast(blob, src=src)
    return blob.to(device), vocab


def broadcast_state_dict(sd, src=0, device="cpu"):
    """broadcast_packed, unpacked into a name -> numpy dict on every rank (for callers that build several contexts from it)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sd
    blob, vocab = broadcast_packed(sd, src=src, device=device)
    return unpack_state_dict(blob.cpu().numpy(), vocab)


def shard_range(n_total, rank, world):
    """Contiguous split of n_total streams: rank r owns [lo, hi)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
