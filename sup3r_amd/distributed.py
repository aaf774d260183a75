"""One-process-per-GPU data parallelism for the Sup3rGan path.

* training: the mini-batch is split on axis 0 into ``world`` equal shards
  (``tf.split`` semantics of ``_get_parallel_grad``,
  sup3r/models/abstract.py:807-841); per-rank gradients are SUMMED with ONE
  RCCL all-reduce over the flat gradient buffer (``s3_params_allreduce_grads``)
  and every rank applies the identical Adam step;
* inference: chunks are sharded over ranks (``ForwardPass.my_chunks``), no
  collective on the data path.

``torch.distributed`` is used only for rendezvous (broadcast of the RCCL unique
id, timing barriers); the gradient collective itself is issued by
libsup3r_hip.so on the context's HIP stream.
"""
import ctypes as C
import os

import numpy as np


def env_rank():
    return (int(os.environ.get('RANK', 0)),
            int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def shard_batch(arr, rank, nranks):
    """Equal split of axis 0 (the batch must divide, like ``tf.split``)."""
    n = arr.shape[0]
    if n % nranks != 0:
        raise ValueError(
            f'batch of {n} observations does not divide over {nranks} GPUs')
    per = n // nranks
    return arr[rank * per:(rank + 1) * per]


def init_data_parallel():
    """Create this rank's Device and (world > 1) its RCCL communicator."""
    import torch
    from . import _lib
    from .engine import Device
    rank, local_rank, world = env_rank()
    torch.cuda.set_device(local_rank)
    dev = Device.get(local_rank)
    if world == 1:
        return dev
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(
            'nccl', device_id=torch.device('cuda', local_rank))
    uid = (C.c_char * 128)()
    if rank == 0:
        _lib.check(_lib.lib().s3_comm_unique_id(uid), None,
                   's3_comm_unique_id')
    payload = [bytes(uid)]
    dist.broadcast_object_list(payload, src=0)
    buf = (C.c_char * 128).from_buffer_copy(payload[0])
    dev.init_comm(rank, world, buf)
    return dev


def sum_over_ranks_host(arrays, group=None):
    """Host-side SUM of a list of numpy arrays over a torch.distributed group
    (gloo).  Test / CPU utility that states the reduction semantics of the
    device collective: elementwise sum, every rank gets the result."""
    import torch
    import torch.distributed as dist
    out = []
    for a in arrays:
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        out.append(t.numpy())
    return out
