"""Multi-GPU plumbing: one process per GPU, samples sharded one per rank (the reference's unit of parallelism is the
sample: src/genome_set.c:66-94).  The only exchange on the path is the all-gatherv of the per-sample histograms (the kept
24-byte records) ahead of the cross-sample merge (reference: src/genome_set.c:195-229,250-289).  torch.distributed is
plumbing here: backend "nccl" is RCCL over xGMI on the GPU node, "gloo" in the CPU tests."""
import numpy as np
import torch

RECORD_BYTES = 24


class _DevMem:
    """zero-copy view of device memory owned by the C library"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes_tensor(ptr, nbytes, device):
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


def all_gatherv_bytes(local, dist, group=None):
    """all-gatherv of a 1-D uint8 tensor: sizes first, then one padded all_gather (xGMI is point-to-point, the payloads
    are MB-scale: one collective on max-padded blocks beats a ring of sends).  Returns (list of tensors, sizes)."""
    world = dist.get_world_size(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":        # rehearsal on a one-GPU box: gloo gathers on the host
        parts, sizes = all_gatherv_bytes(local.cpu(), dist, group)
        return [p.to(local.device) for p in parts], sizes
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes_t = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes_t, n, group=group)
    sizes = [int(x) for x in sizes_t.tolist()]              # the one host synchronisation of the exchange
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=torch.uint8, device=local.device)
    pad[: local.numel()] = local
    out = torch.empty(world * mx, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx: r * mx + sz] for r, sz in enumerate(sizes)], sizes


def all_gather_histograms(counter, dist, group=None):
    """Gather every rank's kept records (device-resident, straight from the library's buffer).
    Returns (uint8 tensor of all records concatenated in rank order, list of record counts)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    n = counter.n_kept
    local = device_bytes_tensor(counter.kept_device_ptr, n * RECORD_BYTES, dev) if n else torch.empty(0, dtype=torch.uint8, device=dev)
    parts, sizes = all_gatherv_bytes(local, dist, group)
    return torch.cat(parts), [s // RECORD_BYTES for s in sizes]


def merge_histograms_device(counter, records_u8, counts):
    """Context-keyed union on the GPU (tjamd_merge_samples): records_u8 = CUDA uint8 tensor holding the samples' kept
    records back to back, counts = records per sample.  Returns (keys uint8 tensor [n_union*24], int32 tensor
    [n_union, n_samples]) in the reference's descending key order."""
    import ctypes as C
    from .capi import lib, TatajubaAmdError, _err
    n = int(sum(counts))
    ns = len(counts)
    keys = torch.empty(max(n, 1) * RECORD_BYTES, dtype=torch.uint8, device=records_u8.device)
    mat = torch.empty((max(n, 1), ns), dtype=torch.int32, device=records_u8.device)
    arr = (C.c_long * ns)(*[int(x) for x in counts])
    torch.cuda.current_stream().synchronize()
    got = lib().tjamd_merge_samples(counter._h, C.c_void_p(records_u8.data_ptr()), arr, ns, C.c_void_p(keys.data_ptr()),
                                    C.c_void_p(mat.data_ptr()), n)
    if got < 0:
        raise TatajubaAmdError(_err())
    return keys[: got * RECORD_BYTES], mat[:got]
