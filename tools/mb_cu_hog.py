"""What a collective in flight costs the training step on the SAME GPU, priced on a one-GPU box: a stand-in kernel (tools/hip/cu_hog.hip) holds
G CUs on a side stream -- the footprint of an RCCL all-reduce's channels -- while the step (hipGraph replay) runs; with and without
DM_GEMM_CUS_RESERVED=G (the one-workgroup-per-CU GEMM grids plan for G CUs fewer).  The hog holds its CUs for the whole step: an upper bound,
an 8-GPU step has a bucket in flight for about a fifth of its time.
    python tools/mb_cu_hog.py            (run once per DM_GEMM_CUS_RESERVED setting: the switch is read once per process)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import bench
from deepmerge_amd.nets.ShfitScaleFormer import ShfitScaleFormer_v3
from deepmerge_amd.trainer import PairTrainer
DEV = "cuda:0"
hog = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "hip", "libcuhog.so"))
hog.cu_hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
scales, in_c, depth, B = [32, 64, 128, 256], 4, [3, 2, 1], 32
torch.manual_seed(0)
net = ShfitScaleFormer_v3(cube_size=[8, 8], input_image_scales=list(scales), depth=list(depth), in_c=in_c, numerics="bf16").to(DEV)
tr = PairTrainer(net, margin=1.0, lr=1e-4); tr.enable_graph(warmup=2)
batch = bench.synth_batch(B, scales, in_c, DEV, 1000)
for _ in range(6): tr.step(*batch)
batch = tr.graph_inputs()
side = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device=DEV)
def run(G, steps=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        if G:
            side.wait_stream(torch.cuda.current_stream())
            hog.cu_hog_launch(side.cuda_stream, G, 5200, sink.data_ptr())          # holds G CUs for ~one step
        tr.step(*batch)
        if G:
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
res = os.environ.get("DM_GEMM_CUS_RESERVED", "0")
print(f"DM_GEMM_CUS_RESERVED={res}: " + "  ".join(f"hog {G:3d} CUs: {run(G):6.3f} ms" for G in (0, 8, 16, 32, 64)), flush=True)
