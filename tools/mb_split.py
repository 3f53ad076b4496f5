import sys, os, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from deepmerge_amd import _lib
lib=_lib.lib(); dev="cuda:0"
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(n)]; b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n*1e-3
st=torch.cuda.current_stream().cuda_stream
for rows,cols in [(16384,3072),(16384,768),(16384,2304),(3072,768)]:
    for stack in (0,1):
        src=torch.randn(rows,cols,device=dev); dst=torch.empty(3*rows*cols,dtype=torch.bfloat16,device=dev)
        s=t(lambda: lib.dm_split_bf16(src.data_ptr(),cols,rows,cols,dst.data_ptr(),stack,0b100,st))
        print(f"split {rows}x{cols} stack={stack}: {s*1e6:7.1f} us  {rows*cols*10/s/1e12:5.2f} TB/s")
