"""Worker for test_device_survivor_exchange_matches_get_results (fresh process): the RCCL transport of
phenotypeseeker_amd.dist on a one-rank communicator -- unique id through the rendezvous file, ncclCommInitRank,
all-reduce, all-gather (host and device buffers), all-to-all."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PSK_RDZV_FILE"] = os.path.join(tempfile.mkdtemp(prefix="psk_rdzv_"), "id")
os.environ.pop("PSK_DIST_TRANSPORT", None)

from phenotypeseeker_amd import dist  # noqa: E402
from phenotypeseeker_amd.engine import PskContext  # noqa: E402

assert "torch" not in sys.modules
g = dist.Group()
g.world, g.rank, g.local_rank = 1, 0, 0
g.init(force=True)
assert g.backend == "rccl" and g.rccl_ranks == 1             # ncclCommCount
assert os.listdir(os.environ["PSK_RDZV_FILE"] + ".rdzv") == []  # id and status files are gone once every rank has joined
assert g.allreduce_sum(41) == 41 and g.allreduce_sum(0.5) == 0.5 and g.allreduce_max(3.0) == 3.0
assert g.allreduce_sum((1 << 63) + 5) == (1 << 63) + 5          # u64, not i64
assert g.allgather_bytes(b"slab") == [b"slab"] and g.allgather_bytes(b"") == [b""]
assert g.allgather_i64(np.array([-7, 9])).tolist() == [[-7, 9]]
g.barrier()
with PskContext(0) as ctx:
    n, m = 200, 300_000
    ctx.synth_presence(m, n, seed=11)
    ph = (np.arange(n) % 2).astype(np.int8)
    npass = ctx.chi2_scan(ph, None, 2, n - 2, 0.05, False, m)
    assert npass > 100, npass
    ref = ctx.get_results(npass)
    ref_bits = ctx.get_rows(ref["row"])
    x = dist.SurvivorExchange(g, ctx.presence_shape()[1], cap_records=64)   # forces a regrow
    assert x.t.stream
    res, bits = x.gather(ctx)
    assert x.cap >= npass
    for key in ("word", "stat", "p", "n_with"):
        assert np.array_equal(res[key], ref[key]), key
    assert np.array_equal(bits, ref_bits)
    s0, _ = x.start(ctx)                      # double-buffered form: two exchanges in flight
    ctx.chi2_scan(1 - ph, None, 2, n - 2, 0.05, False, m)
    s1, _ = x.start(ctx)
    a, b = x.finish(s0), x.finish(s1)
    assert np.array_equal(a[0]["word"], ref["word"]) and np.array_equal(b[0]["word"], ref["word"])
    # the multi-GPU step as bench.py runs it: two scans in flight, export + all-gather of scan i under scan i + 1,
    # alternating phenotypes so that a mixed-up result set would show
    phs = [ph, (np.arange(n) % 3 == 0).astype(np.int8)]
    want = []
    for q in phs:
        c = ctx.chi2_scan(q, None, 2, n - 2, 0.05, False, m)
        want.append(ctx.get_results(c)["word"])
    assert not np.array_equal(want[0], want[1])
    steps, slots = 9, []
    ctx.chi2_scan_begin(phs[0], None, 2, n - 2, 0.05, False, m)
    ctx.chi2_scan_begin(phs[1], None, 2, n - 2, 0.05, False, m)
    for i in range(steps):
        ctx.scan_end()
        s = x.export(ctx)
        if i + 2 < steps:
            ctx.chi2_scan_begin(phs[i % 2], None, 2, n - 2, 0.05, False, m)
        x.collect(s)
        got = x.finish(s)
        assert np.array_equal(got[0]["word"], want[i % 2]), i
# the ingest exchange over RCCL (one rank: the all-to-all is a device-to-device copy through ncclSend / ncclRecv): lists counted in
# one context, moved, installed in another -> the same lists, the same matrix
from phenotypeseeker_amd.synth import GenomeSet  # noqa: E402
gs = GenomeSet(5, 40_000, seed=3, gene_len=300)
datas = [gs.sample(i)[1] for i in range(5)]
with PskContext(0) as cnt, PskContext(0) as slab:
    cnt.begin(11, 5)
    nu, nt = cnt.count_kmers_batch(0, datas, 2)
    slab.begin(11, 5)
    counted = [cnt.get_list(i, nu[i]) for i in range(5)]
    m_cnt, union_cnt = cnt.build_presence(), cnt.get_union()
    pairs = dist.ListExchange(g, 11).run(cnt, slab, 5, nt)      # (releases the lists of cnt)
    assert pairs == sum(nu)
    for i in range(5):
        a, b = counted[i], slab.get_list(i, nu[i])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), i
    assert m_cnt == slab.build_presence()
    assert np.array_equal(union_cnt, slab.get_union())
assert "torch" not in sys.modules
g.close()
print("exchange ok", npass)
