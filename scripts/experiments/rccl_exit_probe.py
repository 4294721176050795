"""Which combination makes the process die with 'double free' at exit after using libmcle's RCCL communicator?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
if "torch" in mode:
    import torch  # noqa: F401
from pyphysim_amd import _lib
from pyphysim_amd.engine import Engine
if "system" in mode:
    _lib.torch_rccl_path = lambda: "/opt/rocm/lib/librccl.so.1"
eng = Engine(0, "f32")
if "nocomm" not in mode:
    from pyphysim_amd.distributed import NativeComm
    comm = NativeComm(eng, rank=0, world=1)
    print(mode, "allreduce:", comm.allreduce_floats([1.0, 2.0]))
    if "noclose" not in mode:
        comm.close()
if "engclose" in mode:
    eng.close()
print(mode, "done", flush=True)
if "late" in mode:
    import torch  # noqa: F401,E402
    import torch.multiprocessing  # noqa: F401,E402
    print(mode, "late torch imported", torch.cuda.is_available(), flush=True)
