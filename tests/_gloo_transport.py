"""Host transport for phenotypeseeker_amd.dist in tests: torch.distributed with the gloo backend stands in for
RCCL where there is no second GPU (the CPU suite; several ranks sharing the one GPU of a gpurun box, where RCCL
refuses two ranks per device).  Selected with PSK_DIST_TRANSPORT=_gloo_transport:GlooTransport (tests/ on
PYTHONPATH).  Not part of the product: the package itself never imports torch.

Buffers handed to the engine are device memory when a GPU is visible (the engine's copies are device-to-device)
and plain host arrays otherwise (the CPU suite's host stand-ins for the contexts)."""
import ctypes
import os

import numpy as np
import torch
import torch.distributed as dist


class _HostBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(max(nbytes, 256))
        self.arr = np.zeros(self.nbytes, dtype=np.uint8)
        self.ptr = self.arr.ctypes.data

    def free(self):
        pass


class GlooTransport:
    name = "gloo"
    stream = 0          # no device stream: exports are the waited-for form

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, device
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        from phenotypeseeker_amd import _lib
        self.ctx = None
        if _lib.load().psk_device_count() > 0:
            from phenotypeseeker_amd.engine import PskContext
            self.ctx = PskContext(device)
        self.device_memory = self.ctx is not None

    # -- buffers ------------------------------------------------------------------------------------
    def alloc(self, nbytes):
        if self.ctx is None:
            return _HostBuffer(nbytes)
        from phenotypeseeker_amd.dist import DeviceBuffer
        return DeviceBuffer(self.ctx, nbytes)

    def to_host(self, buf, nbytes, offset=0):
        if self.ctx is None:
            return buf.arr[offset:offset + nbytes].copy()
        return self.ctx.dev_download(buf.ptr + offset, nbytes)

    def _upload(self, buf, arr, offset=0):
        a = np.ascontiguousarray(arr).view(np.uint8).ravel()
        if a.size == 0:
            return
        if self.ctx is None:
            buf.arr[offset:offset + a.size] = a
        else:
            self.ctx.dev_upload(buf.ptr + offset, a)

    # -- collectives --------------------------------------------------------------------------------
    def allreduce(self, arr, op):
        if arr.dtype == np.uint64:
            t = torch.from_numpy(arr.view(np.int64))
        else:
            t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return arr

    def allgather_host(self, send_u8):
        send = torch.from_numpy(np.ascontiguousarray(send_u8).view(np.uint8).ravel().copy())
        outs = [torch.zeros_like(send) for _ in range(self.world)]
        dist.all_gather(outs, send)
        return np.stack([o.numpy() for o in outs])

    def allgather_device(self, send, recv, nbytes):
        got = self.allgather_host(self.to_host(send, nbytes))
        self._upload(recv, got)

    def alltoallv(self, send, send_counts, recv, recv_counts, elem_bytes):
        sc = [int(c) * elem_bytes for c in send_counts]
        rc = [int(c) * elem_bytes for c in recv_counts]
        src = torch.from_numpy(self.to_host(send, sum(sc)).copy()) if sum(sc) else torch.zeros(0, dtype=torch.uint8)
        dst = torch.zeros(sum(rc), dtype=torch.uint8)
        dist.all_to_all_single(dst, src, rc, sc)
        self._upload(recv, dst.numpy())

    def sync(self):
        pass

    def barrier(self):
        dist.barrier()

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None
        if dist.is_initialized():
            dist.destroy_process_group()
