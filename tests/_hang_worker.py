"""Worker for test_launch_deadline_*: a rank of a launch whose collective never returns.  Uses the product's own pieces --
modeling.Phases (enter / mark / snapshot), watchdog.install, dist.Group with a transport whose all-reduce blocks inside a
system call (the GIL released, as inside libpsk.so's ncclAllReduce + stream wait) -- so that what is under test is what a
hung `phenotypeseeker modeling` rank would run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from phenotypeseeker_amd import dist, watchdog  # noqa: E402
from phenotypeseeker_amd.modeling import Phases  # noqa: E402


class NeverReturns:
    name = "test-transport"
    n_ranks = 0

    def __init__(self):
        self.r, self.w = os.pipe()

    def allreduce(self, arr, op):
        os.read(self.r, 1)          # nobody writes: blocks in read(2) for ever (signals interrupt and re-enter it)
        return arr

    def barrier(self):
        pass

    def close(self):
        pass


def main():
    mode = sys.argv[1]
    ph = Phases()
    watchdog.install(os.environ["RANK"], os.environ["WORLD_SIZE"], ph.snapshot)
    ph.enter("arguments, data.pheno")
    ph.mark("arguments, data.pheno")
    grp = dist.Group()
    grp.init(transport=NeverReturns())
    ph.enter("presence matrix")
    ph.mark("presence matrix")
    if mode == "rank1-hangs" and grp.rank == 0:
        print("rank 0 done")
        return
    ph.enter("all-reduce of the union size")
    grp.allreduce_sum(12345)
    ph.mark("all-reduce of the union size")


if __name__ == "__main__":
    main()
