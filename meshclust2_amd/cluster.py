"""Mean-shift clustering over one GPU PER RANK (SURVEY.md 8(e) + 8(f1); BASELINE cfg3 / cfg5 shape): a launcher.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         -m meshclust2_amd.cluster input.fa --recover weights.txt [--id 0.9] [--kmer K] [--datatype 8|16|32|64] [--sparse] [--output out.clstr]

The work is host C++ (north_star: "host code stays C++ calling HIP through a thin C-ABI"): every rank starts
meshclust2_amd/host/msc_cluster with the RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT it was given. msc_cluster then
  - holds 1 / WORLD_SIZE of the points (block-cyclic in blocks of 1000, one bvec bin: cluster/CRunner.cpp:585) on GPU LOCAL_RANK,
  - runs the clustering logic of host/msc_driver.hpp on replicated bookkeeping (ClusterFactory::MS / accumulate / mean_shift_update /
    merge, cluster/ClusterFactory.cpp:288-435,553-656), so no bookkeeping is ever exchanged,
  - answers its operators through host/msc_sharded.hpp: the query of a get_close step is broadcast as one packed slot and one
    fixed-size record per rank comes back (<= 2 collectives per step), get_mean is a column-sum reduction, an update round costs
    collectives per chunk of centres; RCCL over xGMI between the ranks (host/msc_comm_rccl.hpp), the ncclUniqueId over a socket.
`torch.distributed.run --no-python .../msc_cluster ...` starts the same thing without this file; r02's Python protocol
(DistributedBackend over torch.distributed, dense slots only) is gone.
"""
import os
import subprocess
import sys

BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "msc_cluster")


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    if not os.path.exists(BIN):
        raise ImportError("%s is missing: run `make -C meshclust2_amd/host` (or __graft_entry__.build())" % BIN)
    # a child process, not an exec: nothing here has touched the GPU, but the launcher's process group should see this rank exit
    return subprocess.call([BIN] + argv, env=os.environ.copy())


if __name__ == "__main__":
    sys.exit(main())
