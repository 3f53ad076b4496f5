"""LDS-DMA fill rate per CU for GEMM-operand-like access patterns (tools/hip/fill_bench.hip)."""
import ctypes, os, sys
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip", "libfillbench.so"))
lib.fill_bench.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
buf = torch.randint(0, 255, (1 << 30,), dtype=torch.uint8, device=dev)
sink = torch.zeros(4096, dtype=torch.int32, device=dev)
def run(panels, rows, pitch, row_bytes, ksteps, depth, pieces, grid):
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: lib.fill_bench(buf.data_ptr(), panels, rows, pitch, row_bytes, ksteps, depth, pieces, grid, sink.data_ptr(), st)
    assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 5 * 1e-3
    byts = grid * ksteps * pieces * 4 * 1024
    print(f"panels={panels:4d} rows={rows} pitch={pitch} row_bytes={row_bytes:4d} depth={depth} pieces={pieces:2d} grid={grid}: {dt*1e6:8.1f} us "
          f"{byts/dt/1e12:6.2f} TB/s {byts/dt/1e9/256:6.1f} GB/s/CU", flush=True)
for panels in (4, 16, 64, 1024):
    for row_bytes in (64, 128, 256, 1024):
        for depth, pieces, grid in ((2, 6, 512), (3, 6, 512), (4, 6, 512), (3, 12, 256), (3, 6, 1024), (3, 6, 256)):
            run(panels, 384, 1536, row_bytes, 1536 // row_bytes * 8, depth, pieces, grid)
