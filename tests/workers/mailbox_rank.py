"""One rank of tests/test_gpu_sharded.py::test_mailbox_exchange_between_processes: python mailbox_rank.py RANK WORLD PORT CALLS.
Ranks share GPU 0 (the box has one); torch.distributed (gloo) only carries the 64-byte mailbox handles and the final barrier.  Every rank can
regenerate every other rank's lists (seeded by rank and call), so each checks the merged answer of every call against numpy on its own."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vectordb_amd as amd  # noqa: E402

rank, world, port, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
x = amd.Exchange.direct(rank, world, device=0)
handles = [None] * world
dist.all_gather_object(handles, x.mailbox_export())
x.mailbox_connect(handles)


def lists(r, c, nq, k):
    g = np.random.default_rng(1000 * c + r)
    d = np.sort(g.random((nq, k), dtype=np.float32), axis=1)
    i = (g.integers(0, 1 << 30, (nq, k)).astype(np.int64) * world + r)      # global ids of shard r
    if c % 5 == 0 and nq > 1:
        i[1, k // 2:] = -1                                                    # a short list
        d[1, k // 2:] = np.inf
    return i, d


stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    for c in range(calls):
        nq, k = ((1, 10), (1, 64), (4, 10), (16, 10), (100, 10), (13, 100))[c % 6]
        i, d = lists(rank, c, nq, k)
        ti, td = torch.from_numpy(i).to(dev), torch.from_numpy(d).to(dev)
        oi = torch.empty((nq, k), dtype=torch.int64, device=dev)
        od = torch.empty((nq, k), dtype=torch.float32, device=dev)
        x.direct_merge(ti, td, oi, od, stream=stream.cuda_stream)
        if c % 7 != 3:            # (most calls are read back at once; some are left in flight so that the next call's pushes overlap this call's merge on a peer)
            stream.synchronize()
            alli = np.concatenate([lists(r, c, nq, k)[0] for r in range(world)], axis=1)
            alld = np.concatenate([lists(r, c, nq, k)[1] for r in range(world)], axis=1)
            alld = np.where(alli < 0, np.inf, alld)
            order = np.lexsort((alli, alld), axis=1) if False else np.stack([np.lexsort((alli[q], alld[q])) for q in range(nq)])
            wi = np.take_along_axis(alli, order, 1)[:, :k]
            wd = np.take_along_axis(alld, order, 1)[:, :k]
            gi, gd = oi.cpu().numpy(), od.cpu().numpy()
            ok = np.array_equal(np.where(np.isinf(wd), -1, wi), np.where(np.isinf(gd), -1, gi)) and np.array_equal(wd, gd)
            if not ok:
                print("MISMATCH rank", rank, "call", c, flush=True)
                sys.exit(3)
stream.synchronize()
t = x.times_us(8)
dist.barrier()
x.close()
print("rank %d ok: %d calls, last push+wait / merge us %s" % (rank, calls, [(round(a, 1), round(b, 1)) for a, b in t[-2:]]), flush=True)
