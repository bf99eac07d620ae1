"""
Test transport: the slot protocol of ibo_amd/multigpu.py (fill_slot / reduce_slots, the same
reduction csrc/comm.hip applies after its ncclAllReduce) over torch.distributed's gloo backend,
so the N>1 logic runs on CPU.  torch is imported only inside the spawned worker processes, never
in the pytest process itself (on the GPU box torch would bring a second HIP runtime and its own
librccl into the process that also loads libibo_hip.so).
"""
import numpy as np


class GlooArgmax(object):
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def argmax(self, val, idx, payload=()):
        import torch
        from ibo_amd.multigpu import fill_slot, reduce_slots
        payload = np.asarray(payload, dtype=float).reshape(-1)
        t = torch.from_numpy(fill_slot(self.world_size, self.rank, val, idx, payload))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return reduce_slots(t.numpy(), self.world_size, len(payload))

    def allreduce_sum(self, buf):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.numpy()

    def barrier(self):
        self.dist.barrier(group=self.group)

    def nranks(self):
        return self.world_size


def init_gloo(rank, world, port):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist
