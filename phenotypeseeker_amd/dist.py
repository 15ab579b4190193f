"""Range sharding of the canonical k-mer word space over the GPUs of one node, and the collectives the sharded
path needs:

  1. balanced slab bounds: quantile cuts of a pilot of the lists (the reference's own chunking is perfectly
     balanced -- round-robin `split -n r/<nt>`, modeling.py:335-342 --, and canonical words min(w, rc(w)) are
     far from uniform over the 2k-bit space: density ~ 2(1 - u), 1.87x imbalanced at 8 uniform slabs);
  2. all-reduce(sum) of the per-slab union sizes -> the global Bonferroni denominator
     (phenotypes.no_kmers_to_analyse, modeling.py:644,:738,:795) BEFORE any filtering;
  3. all-gather(v) of each slab's surviving rows (word, statistic, p, n_with, presence bits);
  4. all-to-all(v) of list ranges for the sample-parallel ingest.

One process per GPU.  The collectives run on RCCL over xGMI, bound directly by libpsk.so (csrc/comm.hip:
ncclCommInitRank / ncclAllReduce / ncclAllGather / ncclSend+ncclRecv on the library's own stream); the 128-byte
unique id travels through a private rendezvous directory of the launch (launch.py, or a per-user 0700 directory keyed
by the launcher process).  No PyTorch anywhere in this package: tests that have no second GPU
inject a host transport (tests/_gloo_transport.py, named by PSK_DIST_TRANSPORT=module:Class).  With world size 1
nothing here touches a communicator.
"""
import atexit
import contextlib
import ctypes
import importlib
import os
import sys
import tempfile
import time

import numpy as np

from .engine import PskContext


# ---- slab bounds ---------------------------------------------------------------------------------------------
def slab_bounds(k, world, rank):
    """UNIFORM contiguous slab [lo, hi) of the 2k-bit word space for `rank`; hi == 0 means "to the end"
    (the psk_begin convention, needed because 4**32 does not fit in u64).  Kept as the fallback when no pilot is
    available; runs use balanced_bounds()."""
    space = 1 << (2 * k)
    if world > space:
        raise ValueError("more ranks (%d) than canonical %d-mer words (%d)" % (world, k, space))
    lo = (space * rank) // world
    hi = (space * (rank + 1)) // world
    if rank == world - 1:
        hi = 0
    return lo, hi


def _legal_bounds(cuts, k, world):
    """cuts: world - 1 ascending candidates -> world + 1 bounds [0, c1, ..., 0] with strictly increasing inner
    entries inside the word space (empty slabs are not allowed by psk_begin)."""
    space = 1 << (2 * k)
    if world > space:
        raise ValueError("more ranks (%d) than canonical %d-mer words (%d)" % (world, k, space))
    out, prev = [0], 0
    for r, c in enumerate(cuts):
        c = int(c)
        c = max(c, prev + 1)
        c = min(c, space - (world - 1 - r))     # leave room for the slabs behind
        out.append(c)
        prev = c
    out.append(0)
    return out


def canonical_cdf_bounds(k, world):
    """Closed-form quantile cuts for uniformly random sequence: for a random k-mer w, P(min(w, rc(w)) >= x) is
    close to (1 - x / 4^k)^2 (w and its reverse complement are nearly independent), so the q-quantile of the
    canonical words sits at 4^k (1 - sqrt(1 - q)).  The default when no list has been counted yet."""
    space = 1 << (2 * k)
    cuts = [int(space * (1.0 - (1.0 - r / world) ** 0.5)) for r in range(1, world)]
    return _legal_bounds(cuts, k, world)


def pilot_points(words, n_points=8192):
    """Evenly spaced quantile points of one sorted list (what a rank contributes to the bounds)."""
    words = np.asarray(words, dtype=np.uint64)
    if len(words) == 0:
        return np.zeros(0, dtype=np.uint64)
    idx = (np.arange(n_points, dtype=np.float64) + 0.5) * (len(words) / n_points)
    return words[np.minimum(idx.astype(np.int64), len(words) - 1)]


def quantile_bounds(points, k, world):
    """points: quantile points of the pilot lists (any order) -> world + 1 bounds cutting them into equal parts.
    No points: the closed form."""
    pts = np.sort(np.asarray(points, dtype=np.uint64))
    if len(pts) < 4 * world:
        return canonical_cdf_bounds(k, world)
    cuts = [int(pts[(len(pts) * r) // world]) for r in range(1, world)]
    return _legal_bounds(cuts, k, world)


def balanced_bounds(group, k, pilot_lists, n_points=8192):
    """The same bounds on every rank: each rank contributes the quantile points of the lists it has in hand
    (sorted word arrays, whole word space); one all-gather; equal-count cuts of the merged points."""
    mine = [pilot_points(w, n_points) for w in pilot_lists if len(w)]
    mine = np.concatenate(mine) if mine else np.zeros(0, dtype=np.uint64)
    blobs = group.allgather_bytes(mine.tobytes())
    pts = np.concatenate([np.frombuffer(b, dtype=np.uint64) for b in blobs]) if blobs else mine
    return quantile_bounds(pts, k, group.world)


# ---- transports ----------------------------------------------------------------------------------------------
class DeviceBuffer:
    """A plain allocation the exchanges pack into / receive into (device memory with the RCCL transport)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(max(nbytes, 256))
        self.ptr = ctx.dev_alloc(self.nbytes)

    def free(self):
        if self.ptr:
            try:
                self.ctx.dev_free(self.ptr)
            finally:
                self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_MAGIC = b"PSKRDZV1"
_rdzv_seq = [0]     # transports this process has formed: every rank forms them in the same order, so the n-th one of a
                    # launch meets under the same names on every rank


def _parent_start_ticks():
    """Start time (clock ticks since boot) of the parent process, from /proc/<ppid>/stat: together with its pid it
    names ONE launcher process for ever -- a later launcher that gets the same pid has another start time."""
    try:
        with open("/proc/%d/stat" % os.getppid(), "rb") as f:
            return f.read().rsplit(b")", 1)[1].split()[19].decode()
    except (OSError, IndexError):
        return "0"


def _private_dir(path):
    """mkdir -p `path` with mode 0700 and refuse a directory that somebody else owns, that others can write, or that is
    a symlink (a multi-user box: nobody may plant ids or status files where the ranks meet)."""
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise RuntimeError("rendezvous directory %s is not a private directory of this user (owner %d, mode %o): set "
                           "PSK_RDZV_DIR to a fresh directory of your own" % (path, st.st_uid, st.st_mode & 0o777))
    return path


def _launcher_wide_tag():
    """A tag for the id / status files of a caller-supplied rendezvous directory when PSK_LAUNCH_NONCE is not given: made only of
    what is THE SAME on every rank of one launch and differs between launches; "" when nothing of the kind exists."""
    env = os.environ
    clean = lambda t: "".join(c if c.isalnum() else "_" for c in t)   # noqa: E731
    if env.get("TORCHELASTIC_RUN_ID") and env.get("TORCHELASTIC_RUN_ID") != "none":
        return clean("te_%s_%s" % (env["TORCHELASTIC_RUN_ID"], env.get("TORCHELASTIC_RESTART_COUNT", "0")))
    if env.get("SLURM_JOB_ID"):
        return clean("sl_%s_%s" % (env["SLURM_JOB_ID"], env.get("SLURM_STEP_ID", "0")))
    if env.get("LOCAL_WORLD_SIZE") and env.get("LOCAL_WORLD_SIZE") == env.get("WORLD_SIZE"):
        return clean("pp_%d_%s" % (os.getppid(), _parent_start_ticks()))
    return ""


def _rendezvous():
    """(directory, nonce) where the ranks of THIS launch meet.
      * PSK_RDZV_DIR + PSK_LAUNCH_NONCE: set by this package's own launcher (launch.spawn_ranks): a mkdtemp directory
        made for the launch, and a random nonce every id blob must carry;
      * PSK_RDZV_FILE (a path the caller promises is fresh for this launch): the directory `<file>.rdzv`;
      * otherwise a per-user 0700 directory under the temp dir, then one per launch keyed by MASTER_ADDR / MASTER_PORT /
        the run id AND the launcher process (pid + start time of the parent all ranks share -- torchrun's agent, a
        test's Popen loop), so that a stale file of an earlier or crashed launch can never be met.
    Ranks that do not share a parent (one `bash -c` or srun wrapper per rank) must be given PSK_RDZV_DIR (or
    PSK_RDZV_FILE) by whoever starts them -- a directory that is fresh for the launch, or any directory plus one
    PSK_LAUNCH_NONCE per launch: the error of the rendezvous timing out says so."""
    nonce = os.environ.get("PSK_LAUNCH_NONCE", "")
    # A directory the CALLER supplies is for ranks that need not share a parent (one srun / `bash -c` wrapper per rank):
    # nothing those ranks could derive from their own process tree is the same on all of them (a nonce made of
    # the parent's pid + start time made such ranks reject each other's files until the timeout).  So there the tag every
    # id / status file carries is PSK_LAUNCH_NONCE as given -- empty when the caller gives none, who then promises that the
    # directory is fresh for this launch (a leftover rd./st. file of a crashed launch under the same names WOULD be read)
    # (with no nonce given, a tag is still derived where something launcher-wide and equal on every rank exists -- the
    # elastic launcher's run id + restart count, Slurm's job + step id, or, when every rank of the job is a child of one launcher on
    # this node (LOCAL_WORLD_SIZE == WORLD_SIZE: torchrun with a fixed PSK_RDZV_DIR, the common case), that parent's pid + start
    # time -- so that a rank never reads the unique id a crashed earlier launch left under the same names; only ranks with none of
    # these keep the empty tag and the promise of a fresh directory)
    if not nonce:
        nonce = _launcher_wide_tag()
    d = os.environ.get("PSK_RDZV_DIR")
    if d:
        return _private_dir(d), nonce
    f = os.environ.get("PSK_RDZV_FILE")
    if f:
        return _private_dir(f + ".rdzv"), nonce
    key = "%s_%s_%s_%d_%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"),
                              os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid(), _parent_start_ticks())
    key = "".join(c if c.isalnum() else "_" for c in key)
    base = _private_dir(os.path.join(tempfile.gettempdir(), "psk_rdzv_u%d" % os.getuid()))
    path = _private_dir(os.path.join(base, key))
    if path not in _keyed_dirs:      # a directory made for this launch alone: gone with the process when it is empty
        _keyed_dirs.add(path)
        atexit.register(_rmdir_quietly, path)
    return path, nonce or key


_keyed_dirs = set()


def _rmdir_quietly(path):
    try:
        os.rmdir(path)
    except OSError:
        pass


def _publish(path, payload):
    """Atomic, exclusive publication: the temporary is created with O_EXCL (nobody's planted file is followed), then
    renamed over the name."""
    tmp = "%s.%d.tmp" % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "wb") as fh:
        fh.write(payload)
    os.replace(tmp, path)


def exchange_unique_id(rank, world, make_id, timeout=300.0, rdzv=None, seq=0):
    """Rank 0 removes whatever an earlier transport of the same name left behind (id, collective files),
    creates the id and publishes it as MAGIC | nonce | id; the others poll for a blob that carries THIS launch's nonce
    (anything else -- a stale or foreign file -- is ignored).  Returns (id, path of the id file): rank 0 unlinks it once
    every rank has joined the communicator, and on failure."""
    d, nonce = rdzv or _rendezvous()
    path = os.path.join(d, "id.%d" % seq)
    nb = nonce.encode()
    if rank == 0:
        import shutil
        for name in os.listdir(d):
            if name == "id.%d" % seq or name == "coll.%d" % seq:     # (status files are written by ranks that may be ahead of rank 0)
                full = os.path.join(d, name)
                shutil.rmtree(full, ignore_errors=True) if os.path.isdir(full) else os.unlink(full)
        uid = make_id()
        _publish(path, _MAGIC + len(nb).to_bytes(2, "little") + nb + len(uid).to_bytes(4, "little") + uid)
        return uid, path
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                blob = f.read()
            if blob[:8] == _MAGIC and len(blob) >= 14:
                ln = int.from_bytes(blob[8:10], "little")
                if blob[10:10 + ln] == nb and len(blob) >= 14 + ln:
                    lu = int.from_bytes(blob[10 + ln:14 + ln], "little")
                    if len(blob) == 14 + ln + lu:
                        return blob[14 + ln:], path
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise RuntimeError("rank %d: no unique id of this launch in %s after %.0f s (is rank 0 running?  Ranks that do "
                               "not share a parent process must be given the same PSK_RDZV_DIR -- fresh for the launch, or "
                               "with the same PSK_LAUNCH_NONCE on every rank)" % (rank, path, timeout))
        time.sleep(0.01)


def _tail_of(path, n_lines=6, max_chars=1500):
    """The last lines of a small text file, on one line; "" when there is none."""
    if not path:
        return ""
    try:
        with open(path, "rb") as f:
            text = f.read()[-8192:].decode(errors="replace")
    except OSError:
        return ""
    lines = []
    for ln in text.splitlines():
        ln = ln.strip()
        if "NCCL WARN" in ln:       # (RCCL prefixes a time stamp, host:pid:tid and its source path: the message is what follows)
            ln = ln[ln.index("NCCL WARN"):]
        if "Could not read node #" in ln:      # (rocm-smi's topology files of GPUs the container does not see: a dozen per call)
            continue
        if ln and (not lines or lines[-1] != ln):
            lines.append(ln)
    return " | ".join(lines[-n_lines:])[-max_chars:]


@contextlib.contextmanager
def _stdout_to_stderr():
    """File descriptor 1 points at stderr while the block runs (C stdio flushed on both sides)."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class RcclTransport:
    """RCCL over xGMI through libpsk.so's own communicator (csrc/comm.hip)."""
    name = "rccl"
    device_memory = True

    def __init__(self, rank, world, device, rdzv=None, seq=0):
        self.rank, self.world, self.device = rank, world, device
        self.ctx = PskContext(device)
        self._rdzv = None
        try:
            from . import watchdog
            # RCCL's own account of a refused bring-up is on its debug channel, which it reads ONCE, at its first call in the
            # process (rank 0: ncclGetUniqueId): unless the caller has set that channel up, warnings go to a file of this rank in
            # the rendezvous directory from before that call on, and a failure quotes its last lines
            dbg, dbg_was = None, os.environ.get("NCCL_DEBUG")
            # ("VERSION" -- the banner only, what this pool's boxes export -- counts as not set up: WARN prints the banner too)
            if (dbg_was is None or dbg_was.upper() == "VERSION") and "NCCL_DEBUG_FILE" not in os.environ:
                where = (rdzv or _rendezvous())[0]
                dbg = os.path.join(where, "nccl_warn.%d.%d.log" % (seq, rank))
                os.environ["NCCL_DEBUG"], os.environ["NCCL_DEBUG_FILE"] = "WARN", dbg
            try:
                with watchdog.blocking("waiting for rank 0's RCCL unique id (rendezvous directory)"):
                    uid, self._rdzv = exchange_unique_id(rank, world, self.ctx.comm_unique_id, rdzv=rdzv, seq=seq)
                with _stdout_to_stderr(), watchdog.blocking("ncclCommInitRank (%d ranks)" % world):
                    # (RCCL prints a version banner on stdout; a caller's stdout may be a protocol)
                    self.ctx.comm_init(uid, rank, world)
            except Exception as e:
                tail = _tail_of(dbg)
                if tail:
                    e.args = ("%s; RCCL said: %s" % (e, tail),) + tuple(e.args[1:])
                raise
            finally:
                if dbg:
                    os.environ.pop("NCCL_DEBUG_FILE", None)
                    if dbg_was is None:
                        os.environ.pop("NCCL_DEBUG", None)
                    else:
                        os.environ["NCCL_DEBUG"] = dbg_was
                    try:
                        os.unlink(dbg)
                    except OSError:
                        pass
            self.stream = self.ctx.comm_stream()
            self.barrier()                  # every rank has joined: the file is no longer needed
            self.n_ranks = self.ctx.comm_size()     # ncclCommCount: what a measurement quotes as "rccl_ranks"
            if self.n_ranks != world:
                raise RuntimeError("ncclCommCount = %d in a world of %d" % (self.n_ranks, world))
        except Exception:
            self.ctx.close()
            raise
        finally:
            if rank == 0 and self._rdzv:    # joined, or failed: either way the id is not left behind
                try:
                    os.unlink(self._rdzv)
                except OSError:
                    pass

    def alloc(self, nbytes):
        return DeviceBuffer(self.ctx, nbytes)

    def to_host(self, buf, nbytes, offset=0):
        return self.ctx.dev_download(buf.ptr + offset, nbytes, behind_collectives=True)

    def allreduce(self, arr, op):
        return self.ctx.comm_allreduce(arr, op)

    def allgather_host(self, send_u8):
        return self.ctx.comm_allgather_host(send_u8, self.world)

    def allgather_device(self, send, recv, nbytes):
        self.ctx.comm_allgather_device(send.ptr, recv.ptr, nbytes)

    def alltoallv(self, send, send_counts, recv, recv_counts, elem_bytes):
        self.ctx.comm_alltoallv_device(send.ptr, send_counts, recv.ptr, recv_counts, elem_bytes)

    def sync(self):
        self.ctx.comm_sync()

    def barrier(self):
        self.ctx.comm_allreduce(np.ones(1, dtype=np.uint64), "sum")

    def close(self):
        self.ctx.close()


class HostFileTransport:
    """The same collectives through files in a directory beside the rendezvous file (ranks of ONE node): what a run
    falls back to -- loudly, and named in `Group.backend` -- when RCCL cannot form the communicator AND the run has
    opted in (PSK_DIST_ALLOW_HOST_FILES=1, or PSK_SHARE_GPU=1: several ranks mapped onto one GPU, which RCCL refuses --
    the one-GPU test boxes); without the opt-in a failed communicator is an error (see _rccl_or_host_files).  Collective
    `seq` of rank r is the file `<seq>.<r>` in a 0700 directory of this launch, published by rename; a rank removes its
    file of collective seq - 2 when it has finished seq - 1 (every rank has then read it).  Device buffers are staged
    through the host.  NOT a performance path: a measurement taken over it says so ("scaling": "invalid ...")."""
    name = "host-files"
    device_memory = True
    stream = 0          # no collective stream: exports are the waited-for form

    def __init__(self, rank, world, device, rdzv=None, seq=0, timeout=600.0):
        self.rank, self.world, self.device, self.timeout = rank, world, device, timeout
        d, _ = rdzv or _rendezvous()
        self.dir = _private_dir(os.path.join(d, "coll.%d" % seq))
        self.seq = 0
        self.ctx = PskContext(device)
        self.barrier()

    def _name(self, seq, r):
        return os.path.join(self.dir, "%d.%d" % (seq, r))

    def _exchange(self, payload):
        """all-gather(v) of one bytes object per rank."""
        seq, self.seq = self.seq, self.seq + 1
        _publish(self._name(seq, self.rank), payload)
        out = []
        t0 = time.time()
        for r in range(self.world):
            if r == self.rank:
                out.append(payload)
                continue
            while True:
                try:
                    with open(self._name(seq, r), "rb") as f:
                        out.append(f.read())
                    break
                except OSError:
                    if time.time() - t0 > self.timeout:
                        raise RuntimeError("rank %d: collective %d timed out waiting for rank %d" % (self.rank, seq, r))
                    time.sleep(0.0005)
        if seq >= 2:
            try:
                os.unlink(self._name(seq - 2, self.rank))
            except OSError:
                pass
        return out

    def alloc(self, nbytes):
        return DeviceBuffer(self.ctx, nbytes)

    def to_host(self, buf, nbytes, offset=0):
        return self.ctx.dev_download(buf.ptr + offset, nbytes)

    def allreduce(self, arr, op):
        got = np.stack([np.frombuffer(b, dtype=arr.dtype) for b in self._exchange(arr.tobytes())])
        arr[...] = (got.sum(axis=0, dtype=arr.dtype) if op == "sum" else got.max(axis=0)).reshape(arr.shape)
        return arr

    def allgather_host(self, send_u8):
        send = np.ascontiguousarray(send_u8).view(np.uint8).ravel()
        return np.stack([np.frombuffer(b, dtype=np.uint8) for b in self._exchange(send.tobytes())])

    def allgather_device(self, send, recv, nbytes):
        got = self.allgather_host(self.to_host(send, nbytes)) if nbytes else None
        if nbytes:
            self.ctx.dev_upload(recv.ptr, got)

    def alltoallv(self, send, send_counts, recv, recv_counts, elem_bytes):
        sc = [int(c) * elem_bytes for c in send_counts]
        src = self.to_host(send, sum(sc)).tobytes() if sum(sc) else b""
        head = np.array(sc, dtype=np.int64).tobytes()
        parts = []
        for r, blob in enumerate(self._exchange(head + src)):
            their = np.frombuffer(blob, dtype=np.int64, count=self.world)
            off = 8 * self.world + int(their[: self.rank].sum())
            parts.append(blob[off: off + int(their[self.rank])])
            assert len(parts[-1]) == int(recv_counts[r]) * elem_bytes
        got = np.frombuffer(b"".join(parts), dtype=np.uint8)
        if got.size:
            self.ctx.dev_upload(recv.ptr, got)

    def sync(self):
        pass

    def barrier(self):
        self._exchange(b"\0")

    def close(self):
        """A rank's last files may be removed only when every rank has read them, i.e. has arrived here too: each
        rank leaves a marker, waits for all of them (or for the directory to be gone: rank 0 cleans up last), and
        only then removes what it wrote.  (Removing them at once let a fast rank take its last file away from
        under a slower one, which then sat out the whole timeout.)"""
        import shutil
        try:
            with open(os.path.join(self.dir, "done.%d" % self.rank), "w"):
                pass
            t0 = time.time()
            while time.time() - t0 < 120.0:
                if not os.path.isdir(self.dir) or all(os.path.exists(os.path.join(self.dir, "done.%d" % r)) for r in range(self.world)):
                    break
                time.sleep(0.002)
            for seq in (self.seq - 2, self.seq - 1):
                if seq >= 0:
                    try:
                        os.unlink(self._name(seq, self.rank))
                    except OSError:
                        pass
            if self.rank == 0:
                shutil.rmtree(self.dir, ignore_errors=True)
        except OSError:
            pass
        self.ctx.close()


def host_files_allowed():
    """The host-file fallback is OPT-IN: PSK_DIST_ALLOW_HOST_FILES=1, or PSK_SHARE_GPU=1 (ranks share GPUs, where RCCL
    cannot work at all).  PSK_DIST_STRICT=1 wins over both."""
    if os.environ.get("PSK_DIST_STRICT") == "1":
        return False
    return os.environ.get("PSK_DIST_ALLOW_HOST_FILES") == "1" or os.environ.get("PSK_SHARE_GPU") == "1"


def _exchange_status(d, prefix, rank, world, text, timeout=600.0, nonce=""):
    """Every rank publishes one short text under `<prefix>.<rank>` and reads everybody's: [text of rank 0, ...].  A status is
    MAGIC | nonce | text, like the id blob: a file that carries another launch's nonce (or none) is a leftover and is
    waited out, not read."""
    tag = _MAGIC + len(nonce.encode()).to_bytes(2, "little") + nonce.encode()
    _publish(os.path.join(d, "%s.%d" % (prefix, rank)), tag + (text or "ok").encode())
    out, t0 = [], time.time()
    for r in range(world):
        while True:
            try:
                with open(os.path.join(d, "%s.%d" % (prefix, r)), "rb") as f:
                    blob = f.read()
                if blob[:len(tag)] != tag:
                    raise OSError("a status file of another launch")
                out.append(blob[len(tag):].decode(errors="replace"))
                break
            except OSError:
                if time.time() - t0 > timeout:
                    raise RuntimeError("rank %d: no status `%s` from rank %d after %.0f s in %s (ranks that do not share a parent "
                                       "process need the same PSK_RDZV_DIR and, unless it is fresh, the same PSK_LAUNCH_NONCE)%s"
                                       % (rank, prefix, r, timeout, d, ("; this rank: " + text) if text else ""))
                time.sleep(0.005)
    return out


# where the time of the last communicator bring-up went (read by modeling.Phases: the first run on several GPUs should say
# how much of its start-up is HIP, how much the status rounds, how much ncclCommInitRank)
init_times = {}


def _rccl_or_host_files(rank, world, device):
    """RCCL.  Two rounds of one status file per rank in the launch's rendezvous directory make every decision a
    decision of ALL ranks: (1) before anybody enters ncclCommInitRank -- which blocks until every rank has called it --
    each rank says whether it has its GPU (a rank without a device would leave the others waiting for ever); (2) after
    it, whether the communicator was formed.  When some rank failed, all ranks raise together: a run whose ranks have
    their own GPUs never continues on anything but RCCL, so a curve measured through /tmp files cannot pass for a
    result -- unless the run has opted into the host-file transport (host_files_allowed), which all ranks
    then take together, with a warning on stderr and the reason in `fallback_reason`."""
    rdzv = _rendezvous()
    seq = _rdzv_seq[0]
    _rdzv_seq[0] += 1
    d = rdzv[0]
    t, err = None, ""
    init_times.clear()
    t0 = time.time()
    try:
        probe = PskContext(device)      # does this rank's GPU exist?
        probe.close()
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    init_times["HIP runtime, context"] = time.time() - t0
    t0 = time.time()
    errs = _exchange_status(d, "rd.%d" % seq, rank, world, err, nonce=rdzv[1])
    init_times["waiting for every rank's GPU probe"] = time.time() - t0
    if all(e == "ok" for e in errs):
        t0 = time.time()
        try:
            t = RcclTransport(rank, world, device, rdzv=rdzv, seq=seq)
        except Exception as e:      # PskError from psk_comm_init, or the rendezvous timing out
            err = "%s: %s" % (type(e).__name__, e)
        init_times["unique id + ncclCommInitRank" + ("" if t is not None else " (refused)")] = time.time() - t0
        t0 = time.time()
        errs = _exchange_status(d, "st.%d" % seq, rank, world, err, nonce=rdzv[1])
        init_times["waiting for every rank's communicator"] = time.time() - t0
    bad = [(r, e) for r, e in enumerate(errs) if e != "ok"]
    if bad:
        if t is not None:
            t.close()
        if not host_files_allowed():
            raise RuntimeError("RCCL communicator not formed (rank %d: %s).  Every rank has its own GPU, so nothing else "
                               "will do; tests on a one-GPU box opt into the host-file transport with PSK_SHARE_GPU=1 or "
                               "PSK_DIST_ALLOW_HOST_FILES=1" % bad[0])
        if rank == 0:
            sys.stderr.write("phenotypeseeker_amd.dist: RCCL communicator not formed (rank %d: %s); collectives go "
                             "through host files in %s (allowed by PSK_SHARE_GPU / PSK_DIST_ALLOW_HOST_FILES)\n"
                             % (bad[0][0], bad[0][1], d))
        t = HostFileTransport(rank, world, device, rdzv=rdzv, seq=seq)
        t.fallback_reason = bad[0][1]
    t.barrier()     # everybody has read every status file
    for prefix in ("rd", "st"):
        try:
            os.unlink(os.path.join(d, "%s.%d.%d" % (prefix, seq, rank)))
        except OSError:
            pass
    return t


class Group:
    """The ranks of one run (or a no-op for one rank).  RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run
    and every MPI-style launcher export them."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = self.local_rank   # GPU index of this rank (see init: PSK_SHARE_GPU)
        self.t = None                   # transport
        self.backend = None
        self.rccl_ranks = 0

    def init(self, transport=None, force=False):
        """transport: None (RCCL, or the class named by PSK_DIST_TRANSPORT=module:Class -- tests), or an
        instance.  PSK_SHARE_GPU=1 maps the ranks onto the visible GPUs modulo their count (tests on a one-GPU
        box; RCCL itself refuses two ranks on one GPU)."""
        if self.world == 1 and not force:
            return self
        spec = os.environ.get("PSK_DIST_TRANSPORT")
        cls = None
        if transport is None and spec:
            mod, _, name = spec.partition(":")
            cls = getattr(importlib.import_module(mod), name)      # before libpsk.so is opened (see the module's note)
        if os.environ.get("PSK_SHARE_GPU") == "1":
            from . import _lib
            self.device = self.local_rank % max(_lib.load().psk_device_count(), 1)
        if transport is None:
            transport = cls(self.rank, self.world, self.device) if cls else _rccl_or_host_files(self.rank, self.world, self.device)
        self.t = transport
        self.backend = transport.name
        self.rccl_ranks = int(getattr(transport, "n_ranks", 0))     # ncclCommCount; 0 unless the transport IS RCCL
        return self

    def _blocking(self, what):
        from . import watchdog
        return watchdog.blocking("%s (%s, %d ranks)" % (what, self.backend, self.world))

    def barrier(self):
        if self.t is not None:
            with self._blocking("barrier"):
                self.t.barrier()

    def close(self):
        if self.t is not None:
            with self._blocking("communicator teardown"):
                self.t.close()
        self.t = None

    # -- collectives on small host values ---------------------------------------------------------
    def allreduce_sum(self, value):
        """Sum of a python int / float over ranks (exact for non-negative ints below 2^64)."""
        if self.t is None:
            return value
        with self._blocking("all-reduce(sum)"):
            if isinstance(value, (int, np.integer)):
                return int(self.t.allreduce(np.array([int(value)], dtype=np.uint64), "sum")[0])
            return float(self.t.allreduce(np.array([float(value)], dtype=np.float64), "sum")[0])

    def allreduce_max(self, value):
        if self.t is None:
            return float(value)
        with self._blocking("all-reduce(max)"):
            return float(self.t.allreduce(np.array([float(value)], dtype=np.float64), "max")[0])

    def allgather_i64(self, arr):
        """Equal-length int64 arrays -> [world, n]."""
        a = np.ascontiguousarray(arr, dtype=np.int64)
        if self.t is None:
            return a[None, :]
        with self._blocking("all-gather of %d bytes per rank" % a.nbytes):
            return self.t.allgather_host(a.view(np.uint8)).view(np.int64).reshape(self.world, -1)

    def allgather_bytes(self, payload):
        """all-gather(v) of one bytes object per rank -> list of bytes in rank order."""
        if self.t is None:
            return [payload]
        sizes = self.allgather_i64(np.array([len(payload)]))[:, 0]
        cap = int(max(int(sizes.max()), 8))
        buf = np.zeros(cap, dtype=np.uint8)
        buf[: len(payload)] = np.frombuffer(payload, dtype=np.uint8)
        with self._blocking("all-gather(v) of up to %d bytes per rank" % cap):
            got = self.t.allgather_host(buf)
        return [got[r, : int(sizes[r])].tobytes() for r in range(self.world)]


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
    """All-gather(v) of the scan survivors without leaving the GPU (RCCL over xGMI): every rank packs its
    survivors into a fixed-capacity device buffer (psk_export_survivors_async, queued on the communicator's
    stream), one ncclAllGather queued behind it moves all slabs, and the buffers are doubled so that the
    collective of scan i overlaps scan i+1 (several phenotypes are scanned back to back).  A host transport
    (tests) stages the packed buffer through host memory instead."""

    def __init__(self, group, words_per_row, cap_records=4096):
        self.g = group
        self.t = group.t
        self.wpr = int(words_per_row)
        self.rec_words = 6 + self.wpr
        self.slot = 0
        self.send = self.recv = None
        self._alloc(int(cap_records))

    def _alloc(self, cap):
        for b in (self.send or []) + (self.recv or []):
            b.free()
        self.cap = cap
        self.nbytes = (cap + 1) * self.rec_words * 8
        self.send = [self.t.alloc(self.nbytes) for _ in range(2)]
        self.recv = [self.t.alloc(self.g.world * self.nbytes) for _ in range(2)]

    def export(self, ctx):
        """First half of an exchange, right after a scan has ended: this rank's survivors are packed into the next
        send buffer.  The pack is queued on the communicator's stream and not waited for (the next scan of `ctx`
        that writes the same result set waits for it on the device), so the caller can launch that scan before
        calling collect()."""
        s = self.slot
        self.slot ^= 1
        if self.t.stream:
            ctx.export_survivors_async(self.send[s].ptr, self.cap, self.t.stream)
        else:
            ctx.export_survivors(self.send[s].ptr, self.cap)
        return s

    def collect(self, s):
        """Second half: queues the all-gather of slot s (ordered behind its export on the device)."""
        self.t.allgather_device(self.send[s], self.recv[s], self.nbytes)

    def start(self, ctx):
        """export() + collect() in one call; returns (slot, None)."""
        s = self.export(ctx)
        self.collect(s)
        return s, None

    def _host_table(self, s):
        with self.g._blocking("waiting for the survivors' all-gather (%d bytes per rank)" % self.nbytes):
            full = self.t.to_host(self.recv[s], self.g.world * self.nbytes)
        return full.view(np.uint64).reshape(self.g.world, self.cap + 1, self.rec_words)

    def finish(self, s):
        """Waits for slot s; returns (res dict, bits) of ALL slabs, ascending by word.  Returns None
        when some rank had more survivors than the buffers hold (caller grows and repeats the scan's
        exchange)."""
        host = self._host_table(s)
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
        """Everything queued later on the communicator's stream (the next export into slot s) is ordered behind
        the collective of slot s by the stream itself: nothing to do on the host."""
        return None

    def finish_counts(self, s):
        """Waits for the collectives queued so far and reads back only the per-slab record counts of slot s (the
        records stay on the device); used where the merged table is not needed on the host right away."""
        with self.g._blocking("waiting for the survivors' all-gather (%d bytes per rank)" % self.nbytes):
            return np.array([int(self.t.to_host(self.recv[s], 8, offset=r * self.nbytes).view(np.uint64)[0])
                             for r in range(self.g.world)], dtype=np.int64)

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
    words and one of the counts (ncclSend / ncclRecv in one group over xGMI; staged through host memory by the
    test transport) -- and installs what it receives as the lists of its slab context (psk_set_lists_device).  The
    lists every rank ends up with are the ones it would have counted itself with the slab filter.
    bounds: the world + 1 slab bounds every rank agreed on (balanced_bounds)."""

    def __init__(self, group, k, bounds=None):
        self.g = group
        self.t = group.t
        self.k = int(k)
        W = group.world
        self.bounds = list(bounds) if bounds is not None else [slab_bounds(self.k, W, d)[0] for d in range(W)] + [0]

    def run(self, cnt_ctx, slab_ctx, n_samples, n_total_own):
        """cnt_ctx holds the lists of this rank's samples (local index j = j-th sample i with owner_of(i) == rank) and
        does not hold them any more when this returns (PskContext.release_lists); slab_ctx has been begun with this
        rank's slab and n_samples.  Returns the number of (word, sample) pairs installed."""
        W, r = self.g.world, self.g.rank
        own = [i for i in range(n_samples) if owner_of(i, W) == r]
        per = (n_samples + W - 1) // W                       # rows of the count table, the same on every rank
        cuts = cnt_ctx.lists_split(0, len(own), self.bounds) if own else np.zeros((0, W + 1), np.int64)
        seg = np.zeros((per, W), dtype=np.int64)             # seg[j][d]: entries of my j-th sample that go to rank d
        tot = np.zeros(per, dtype=np.int64)
        if own:
            seg[: len(own)] = np.diff(cuts, axis=1)
            tot[: len(own)] = np.asarray(n_total_own, dtype=np.int64)
        tables = self.g.allgather_i64(np.concatenate([seg.ravel(), tot]))
        seg_all = [t[: per * W].reshape(per, W) for t in tables]     # seg_all[src][j][dst]
        tot_all = [t[per * W:] for t in tables]
        send_counts = [int(seg[:, d].sum()) for d in range(W)]
        recv_counts = [int(seg_all[src][:, r].sum()) for src in range(W)]
        n_send, n_recv = sum(send_counts), sum(recv_counts)
        bufs = []
        try:
            send_w, send_f = self.t.alloc(n_send * 8), self.t.alloc(n_send * 4)
            bufs += [send_w, send_f]
            # destination-major, my samples in order inside: one packing call (waited for inside)
            rs = [(j, int(cuts[j, d]), int(seg[j, d])) for d in range(W) for j in range(len(own))]
            if rs:
                cnt_ctx.copy_list_ranges([x[0] for x in rs], [x[1] for x in rs], [x[2] for x in rs], send_w.ptr, send_f.ptr)
            # packed, the counting context's lists are dead: they go back to the device BEFORE the receive buffers are
            # taken.  At N = 2 of config 3 a rank holds 61 GB of lists; lists + send + receive + the slab context's copy
            # would stand at 244 GB of the 288, without them the peak is 183 GB.  (cnt_ctx is the caller's to close.)
            release = getattr(cnt_ctx, "release_lists", None)
            if release is not None:
                release()
            recv_w, recv_f = self.t.alloc(n_recv * 8), self.t.alloc(n_recv * 4)
            bufs += [recv_w, recv_f]
            # what arrives is source-major, the source's samples in order inside: one installing call
            idx, cnt, tot = [], [], []
            for src in range(W):
                theirs = [i for i in range(n_samples) if owner_of(i, W) == src]
                idx += theirs
                cnt += [int(seg_all[src][j, r]) for j in range(len(theirs))]
                tot += [int(tot_all[src][j]) for j in range(len(theirs))]
            with self.g._blocking("all-to-all(v) of the lists: %d pairs out, %d in" % (n_send, n_recv)):
                self.t.alltoallv(send_w, send_counts, recv_w, recv_counts, 8)
                self.t.alltoallv(send_f, send_counts, recv_f, recv_counts, 4)
                slab_ctx.set_lists_device(idx, cnt, tot, recv_w.ptr, recv_f.ptr)     # (waits for the collectives' stream)
        finally:
            for b in bufs:
                b.free()
        return int(sum(cnt))
