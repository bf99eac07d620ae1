import os, sys, ctypes
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
if mode == "torch_first":
    import torch.distributed as dist
from ibo_amd import _lib
from ibo_amd.multigpu import RcclArgmax
print("devices", _lib.device_count(), {k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "ROCR" in k})
if mode == "hip_first":
    _lib.check(_lib.lib.ibo_device_synchronize(0))
uid = RcclArgmax.unique_id()
print("uid ok")
c = RcclArgmax(1, 0, uid, device=0)
print(mode, "comm ok", c.argmax(1.0, 5, [2.0]))
os.system("grep -E 'hip|rccl' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
