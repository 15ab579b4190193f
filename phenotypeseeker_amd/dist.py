"""Range sharding of the canonical k-mer word space over the GPUs of one node, and the two small
collectives the sharded path needs:

  1. all-reduce(sum) of the per-slab union sizes -> the global Bonferroni denominator
     (phenotypes.no_kmers_to_analyse, modeling.py:644,:738,:795) BEFORE any filtering;
  2. all-gather(v) of each slab's surviving rows (word, statistic, p, n_with, presence bits).

One process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  torch is plumbing here: the data path never leaves libpsk.so.  With
world size 1 nothing in this module touches torch.
"""
import os

import numpy as np


def slab_bounds(k, world, rank):
    """Contiguous slab [lo, hi) of the 2k-bit word space for `rank`; hi == 0 means "to the end"
    (the psk_begin convention, needed because 4**32 does not fit in u64).  Concatenating the
    slabs in rank order reproduces glistmaker's ascending list order."""
    space = 1 << (2 * k)
    if world > space:
        raise ValueError("more ranks (%d) than canonical %d-mer words (%d)" % (world, k, space))
    lo = (space * rank) // world
    hi = (space * (rank + 1)) // world
    if rank == world - 1:
        hi = 0
    return lo, hi


class Group:
    """Thin view of the default torch.distributed process group (or a no-op for one rank)."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = self.local_rank   # GPU index of this rank (see init: PSK_SHARE_GPU)
        self._dist = None
        self._dev = None

    def init(self, backend=None, force=False):
        if self.world == 1 and not force:
            return self
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # test knobs for a box with fewer GPUs than ranks: PSK_DIST_BACKEND=gloo keeps the collectives on the host,
        # PSK_SHARE_GPU=1 maps the ranks onto the visible GPUs modulo their count
        if backend is None:
            backend = os.environ.get("PSK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("PSK_SHARE_GPU") == "1":
            self.device = self.local_rank % max(torch.cuda.device_count(), 1)
        if backend == "nccl":
            torch.cuda.set_device(self.device)
            self._dev = torch.device("cuda", self.device)
        else:
            self._dev = torch.device("cpu")
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
        self._dist = dist
        self.backend = backend
        return self

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
        self._dist = None

    # -- collectives on small host arrays ---------------------------------------------------------
    def allreduce_sum(self, value):
        """Sum of a python int / float over ranks (exact for ints below 2^63)."""
        if self._dist is None:
            return value
        import torch
        is_int = isinstance(value, (int, np.integer))
        t = torch.tensor([int(value) if is_int else float(value)],
                         dtype=torch.int64 if is_int else torch.float64, device=self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return int(t.item()) if is_int else float(t.item())

    def allreduce_max(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def allgather_bytes(self, payload):
        """all-gather(v) of one bytes object per rank -> list of bytes in rank order."""
        if self._dist is None:
            return [payload]
        import torch
        n = torch.tensor([len(payload)], dtype=torch.int64, device=self._dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=self._dev) for _ in range(self.world)]
        self._dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        cap = max(max(sizes), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=self._dev)
        if len(payload):
            buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(self._dev)
        outs = [torch.zeros(cap, dtype=torch.uint8, device=self._dev) for _ in range(self.world)]
        self._dist.all_gather(outs, buf)
        return [bytes(o[:s].cpu().numpy().tobytes()) for o, s in zip(outs, sizes)]


_FIELDS = (("word", np.uint64), ("stat", np.float64), ("p", np.float64), ("mean_x", np.float64),
           ("mean_y", np.float64), ("n_with", np.int32))


def pack_candidates(res, bits):
    """One slab's surviving rows -> bytes: header (n, wpr) + the SoA columns + presence rows."""
    n = len(res["word"])
    wpr = bits.shape[1] if n else 0
    parts = [np.array([n, wpr], dtype=np.int64).tobytes()]
    for name, dt in _FIELDS:
        parts.append(np.ascontiguousarray(res[name], dtype=dt).tobytes())
    parts.append(np.ascontiguousarray(bits, dtype=np.uint64).tobytes())
    return b"".join(parts)


def unpack_candidates(payload):
    n, wpr = (int(x) for x in np.frombuffer(payload, dtype=np.int64, count=2))
    off = 16
    res = {}
    for name, dt in _FIELDS:
        sz = n * np.dtype(dt).itemsize
        res[name] = np.frombuffer(payload, dtype=dt, count=n, offset=off).copy()
        off += sz
    bits = np.frombuffer(payload, dtype=np.uint64, count=n * wpr, offset=off).reshape(n, wpr).copy() if n else \
        np.zeros((0, wpr), dtype=np.uint64)
    return res, bits


def merge_candidates(payloads):
    """Concatenate the slabs in rank order: slabs are ascending ranges, so the result is in
    ascending word order -- byte-identical to what one GPU produces for the whole space."""
    parts = [unpack_candidates(p) for p in payloads]
    res = {name: np.concatenate([p[0][name] for p in parts]) for name, _ in _FIELDS}
    wpr = max([p[1].shape[1] for p in parts] + [0])
    bits = np.concatenate([p[1] if p[1].shape[1] == wpr else np.zeros((0, wpr), np.uint64) for p in parts]) \
        if parts else np.zeros((0, wpr), np.uint64)
    return res, bits


class SurvivorExchange:
    """All-gather(v) of the scan survivors without leaving the GPU (RCCL over xGMI): every rank packs
    its survivors into a fixed-capacity device buffer (psk_export_survivors), one
    all_gather_into_tensor moves all slabs, and the copy is double-buffered so that the collective of
    scan i overlaps scan i+1 (several phenotypes are scanned back to back).  With the gloo backend
    (CPU tests, one-GPU dry runs) the packed buffer is staged through host tensors instead."""

    def __init__(self, group, words_per_row, cap_records=4096):
        import torch
        self.g = group
        self.wpr = int(words_per_row)
        self.rec_words = 6 + self.wpr
        self.torch = torch
        self.slot = 0
        self.work = [None, None]
        self._alloc(int(cap_records))

    def _alloc(self, cap):
        torch = self.torch
        self.cap = cap
        n = (cap + 1) * self.rec_words
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.dev = dev
        self.send = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(2)]
        nccl = getattr(self.g, "backend", None) == "nccl"
        rdev = dev if nccl else torch.device("cpu")
        self.recv = [torch.zeros(self.g.world * n, dtype=torch.int64, device=rdev) for _ in range(2)]
        self.nccl = nccl

    def export(self, ctx):
        """First half of an exchange, right after a scan has ended: this rank's survivors are packed into the next
        send buffer.  With nccl the pack is queued on torch's current stream and not waited for (the next scan of
        `ctx` waits for it on the device), so the caller can launch that scan before calling collect()."""
        s = self.slot
        self.slot ^= 1
        if self.nccl:
            ctx.export_survivors_async(self.send[s].data_ptr(), self.cap, self.torch.cuda.current_stream().cuda_stream)
        else:
            ctx.export_survivors(self.send[s].data_ptr(), self.cap)
        return s

    def collect(self, s):
        """Second half: queues the all-gather of slot s (ordered behind its export on the device)."""
        dist = self.g._dist
        if self.nccl:
            self.work[s] = dist.all_gather_into_tensor(self.recv[s], self.send[s], async_op=True)
        else:
            parts = list(self.recv[s].chunk(self.g.world))
            self.work[s] = dist.all_gather(parts, self.send[s].cpu(), async_op=True)

    def start(self, ctx):
        """export() + collect() in one call; returns (slot, None)."""
        s = self.export(ctx)
        self.collect(s)
        return s, None

    def finish(self, s):
        """Waits for slot s; returns (res dict, bits) of ALL slabs, ascending by word.  Returns None
        when some rank had more survivors than the buffers hold (caller grows and repeats the scan's
        exchange)."""
        self.work[s].wait()
        host = self.recv[s].cpu().numpy().view(np.uint64).reshape(self.g.world, self.cap + 1, self.rec_words)
        counts = host[:, 0, 0].astype(np.int64)
        if (counts > self.cap).any():
            return None
        recs = np.concatenate([host[r, 1:1 + counts[r]] for r in range(self.g.world)]) if counts.sum() else \
            np.zeros((0, self.rec_words), dtype=np.uint64)
        order = np.argsort(recs[:, 0], kind="stable")
        recs = recs[order]
        res = {"word": recs[:, 0].copy(), "stat": recs[:, 1].copy().view(np.float64), "p": recs[:, 2].copy().view(np.float64),
               "mean_x": recs[:, 3].copy().view(np.float64), "mean_y": recs[:, 4].copy().view(np.float64),
               "n_with": recs[:, 5].copy().view(np.int64).astype(np.int32)}
        return res, np.ascontiguousarray(recs[:, 6:])

    def wait(self, s):
        """Orders everything queued later behind the collective of slot s, so that its buffers may be packed again.
        With nccl that is a stream-level wait (the host does not block); gloo completes on the host."""
        self.work[s].wait()

    def finish_counts(self, s):
        """Waits for slot s and reads back only the per-slab record counts (the records stay on the
        device); used where the merged table is not needed on the host right away."""
        self.work[s].wait()
        hdr = self.recv[s].view(self.g.world, self.cap + 1, self.rec_words)[:, 0, 0]
        return hdr.cpu().numpy().astype(np.int64)

    def gather(self, ctx):
        """Synchronous form used by the pipeline: exchange the survivors of the last scan, growing the
        buffers if any slab overflowed."""
        while True:
            s, _ = self.start(ctx)
            out = self.finish(s)
            if out is not None:
                return out
            self._alloc(self.cap * 4)


def owner_of(sample_idx, world):
    """Rank that counts sample `sample_idx` in the sample-parallel ingest (round robin: neighbouring samples, which
    tend to be of similar size, spread over the ranks)."""
    return sample_idx % world


class ListExchange:
    """Multi-GPU ingest without redundant counting: every sample is counted, unfiltered, on ONE rank (context
    `cnt_ctx`, samples numbered locally in ascending global order); a slab of the word space is a contiguous range
    of the sorted list, so each rank then sends range s of each of its lists to rank s -- one all-to-all(v) of the
    words and one of the counts (RCCL over xGMI with nccl; staged through host tensors with gloo) -- and installs
    what it receives as the lists of its slab context (psk_set_lists_device).  The lists every rank ends up with are
    the ones it would have counted itself with the slab filter."""

    def __init__(self, group, k):
        import torch
        self.g = group
        self.torch = torch
        self.k = int(k)
        self.nccl = getattr(group, "backend", None) == "nccl"
        self.dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def _all_gather_i64(self, arr):
        torch, dist = self.torch, self.g._dist
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(self.g._dev)
        outs = [torch.zeros_like(t) for _ in range(self.g.world)]
        dist.all_gather(outs, t)
        return [o.cpu().numpy() for o in outs]

    def run(self, cnt_ctx, slab_ctx, n_samples, n_total_own):
        """cnt_ctx holds the lists of this rank's samples (local index j = j-th sample i with owner_of(i) == rank);
        slab_ctx has been begun with this rank's slab and n_samples.  Returns the number of (word, sample) pairs
        installed."""
        torch, dist = self.torch, self.g._dist
        W, r = self.g.world, self.g.rank
        own = [i for i in range(n_samples) if owner_of(i, W) == r]
        per = (n_samples + W - 1) // W                       # rows of the count table, the same on every rank
        bounds = [slab_bounds(self.k, W, d)[0] for d in range(W)] + [0]
        cuts = cnt_ctx.lists_split(0, len(own), bounds) if own else np.zeros((0, W + 1), np.int64)
        seg = np.zeros((per, W), dtype=np.int64)             # seg[j][d]: entries of my j-th sample that go to rank d
        tot = np.zeros(per, dtype=np.int64)
        if own:
            seg[: len(own)] = np.diff(cuts, axis=1)
            tot[: len(own)] = np.asarray(n_total_own, dtype=np.int64)
        tables = self._all_gather_i64(np.concatenate([seg.ravel(), tot]))
        seg_all = [t[: per * W].reshape(per, W) for t in tables]     # seg_all[src][j][dst]
        tot_all = [t[per * W:] for t in tables]
        send_counts = [int(seg[:, d].sum()) for d in range(W)]
        recv_counts = [int(seg_all[src][:, r].sum()) for src in range(W)]
        stage = self.dev if torch.cuda.is_available() else torch.device("cpu")
        send_w = torch.empty(max(sum(send_counts), 1), dtype=torch.int64, device=stage)
        send_f = torch.empty(max(sum(send_counts), 1), dtype=torch.int32, device=stage)
        # destination-major, my samples in order inside: one packing call
        rs = [(j, int(cuts[j, d]), int(seg[j, d])) for d in range(W) for j in range(len(own))]
        if rs:
            cnt_ctx.copy_list_ranges([x[0] for x in rs], [x[1] for x in rs], [x[2] for x in rs], send_w.data_ptr(),
                                     send_f.data_ptr())
        n_recv = sum(recv_counts)
        if self.nccl:
            recv_w = torch.empty(max(n_recv, 1), dtype=torch.int64, device=stage)
            recv_f = torch.empty(max(n_recv, 1), dtype=torch.int32, device=stage)
            torch.cuda.synchronize()
            dist.all_to_all_single(recv_w[:n_recv], send_w[: sum(send_counts)], recv_counts, send_counts)
            dist.all_to_all_single(recv_f[:n_recv], send_f[: sum(send_counts)], recv_counts, send_counts)
            torch.cuda.synchronize()
        else:                                                # host-staged collectives, device buffers either side
            hw = torch.empty(max(n_recv, 1), dtype=torch.int64)
            hf = torch.empty(max(n_recv, 1), dtype=torch.int32)
            dist.all_to_all_single(hw[:n_recv], send_w[: sum(send_counts)].cpu(), recv_counts, send_counts)
            dist.all_to_all_single(hf[:n_recv], send_f[: sum(send_counts)].cpu(), recv_counts, send_counts)
            recv_w, recv_f = hw.to(stage), hf.to(stage)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        # what arrived is source-major, the source's samples in order inside: one installing call
        idx, cnt, tot = [], [], []
        for src in range(W):
            theirs = [i for i in range(n_samples) if owner_of(i, W) == src]
            idx += theirs
            cnt += [int(seg_all[src][j, r]) for j in range(len(theirs))]
            tot += [int(tot_all[src][j]) for j in range(len(theirs))]
        slab_ctx.set_lists_device(idx, cnt, tot, recv_w.data_ptr(), recv_f.data_ptr())
        return int(sum(cnt))
