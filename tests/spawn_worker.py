"""Child of tests/test_distributed_cpu.py::test_launch_ranks_*: stands in for ``gdn_amd.GDN_main`` under
``distributed.launch_ranks`` (the reference's ``--gpu_num 0,1,2,3`` idiom).  Joins the process group the launcher's
environment describes (gloo here: no GPU), all-reduces its rank and writes what it saw."""
import json
import os
import sys

import torch


def main(argv):
    out_dir, fail_rank = argv[0], int(argv[1])
    from gdn_amd import distributed as D
    rank, local_rank, world = D.init()
    assert os.environ.get("GDN_SPAWNED") == "1"
    if rank == fail_rank:
        sys.exit(7)                       # a rank dying before the collective: the launcher must stop the others
    t = torch.tensor([float(rank + 1)])
    for w in D.allreduce_flat(t):
        w.wait()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "local_rank": local_rank, "world": world, "sum": float(t),
                   "visible": os.environ.get("HIP_VISIBLE_DEVICES")}, f)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
