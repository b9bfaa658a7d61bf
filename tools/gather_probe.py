"""Single-rank probe of the exchange step's cost: RCCL gather vs all_gather vs place kernel."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": dev} if os.environ.get("PROBE_DEVICE_ID") else {}))

def timeit(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: issue {1e6*(t1-t0)/n:.1f} us/call, complete {1e6*(t2-t0)/n:.1f} us/call", flush=True)

for rows in (1000, 125):
    send = torch.zeros((rows, 1000), dtype=torch.int32, device=dev)
    recv_all = torch.empty((1, rows, 1000), dtype=torch.int32, device=dev)
    recv = [recv_all[0]]
    timeit(f"gather {rows}x1000", lambda: dist.gather(send, recv, dst=0))
    timeit(f"all_gather_into_tensor {rows}x1000", lambda: dist.all_gather_into_tensor(recv_all, send))
    timeit(f"copy_ {rows}x1000", lambda: recv_all[0].copy_(send))
from raytracers_amd.dist import HipPartRenderer
pr = HipPartRenderer("rgbbox", 1000, 1000, dev)
img = torch.empty((1000, 1000), dtype=torch.int32, device=dev)
stacked = torch.zeros((1, 1000, 1000), dtype=torch.int32, device=dev)
timeit("place_all 1000x1000", lambda: pr.place_all(1, 1000, stacked, img))
timeit("render rgbbox", lambda: pr(0, 1, img))
send = torch.zeros((1000, 1000), dtype=torch.int32, device=dev)
recv_all = torch.empty((1, 1000, 1000), dtype=torch.int32, device=dev)
recv = [recv_all[0]]
def frame():
    pr(0, 1, send)
    dist.gather(send, recv, dst=0)
    pr.place_all(1, 1000, recv_all, img)
timeit("render+gather+place", frame)
def frame2():
    pr(0, 1, send)
    recv_all[0].copy_(send)
    pr.place_all(1, 1000, recv_all, img)
timeit("render+copy_+place", frame2)
def frame3():
    pr(0, 1, send)
    dist.gather(send, recv, dst=0)
timeit("render+gather", frame3)
dist.destroy_process_group()
