import sys, time, torch
sys.path.insert(0, "/root/repo")
import vectordb_amd as amd
n, d = int(sys.argv[1]), int(sys.argv[2])
g = torch.Generator(device="cuda").manual_seed(42)
X = torch.rand((n, d), generator=g, device="cuda")
ix = amd.GpuIndex(d, 0).use_torch_stream()
ix.attach_rows(X)
torch.cuda.synchronize(); t0 = time.perf_counter()
ix.build()
torch.cuda.synchronize(); print("build_s", time.perf_counter() - t0, ix.graph_info())
