import sys, importlib, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); st = importlib.import_module("pem_spgemm_amd.standins")
rows, cols, I, J, V = st.make("webbase-1M")
ctx = pkg.Context(0)
A = pkg.Tiled.from_coo(ctx, rows, cols, I, J, V)
plan = pkg.CPlan(ctx, A, A); plan.spgemm()
info = plan.info(); nz = info["nnz_c"]; nrows = info["row_end"]-info["row_begin"]
dev = torch.device("cuda", 0)
rp = torch.empty(nrows+1, dtype=torch.int32, device=dev); ci = torch.empty(nz, dtype=torch.int32, device=dev); v = torch.empty(nz, dtype=torch.float64, device=dev)
ctx.set_kernel_profiling(True); ctx.reset_kernel_stats()
for _ in range(3):
    torch.cuda.synchronize(); t=time.perf_counter()
    plan.export_csr_device(rp.data_ptr(), ci.data_ptr(), v.data_ptr()); ctx.synchronize()
    print("export ms", (time.perf_counter()-t)*1e3)
for k,vv in ctx.kernel_stats().items(): print(k, vv)
