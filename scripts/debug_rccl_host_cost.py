"""Does initialising RCCL (torch.distributed, backend nccl) change the host cost of a kernel launch in this process?"""
import os, time, torch, torch.distributed as dist
torch.cuda.set_device(0)
y = torch.ones(1000, device="cuda")
def tiny(n=3000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): y.add_(1.0)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print(f"tiny torch op before RCCL init: {tiny():.1f} us, again {tiny():.1f} us")
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29977")
dist.init_process_group("nccl", rank=0, world_size=1)
print(f"after init_process_group (no collective yet): {tiny():.1f} us")
x = torch.ones(4_000_000, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()
print(f"after the first all_reduce: {tiny():.1f} us, again {tiny():.1f} us")
import threading
print("threads:", threading.active_count(), "os threads:", len(os.listdir(f"/proc/{os.getpid()}/task")))
g2 = dist.new_group(); dist.all_reduce(x, group=g2); torch.cuda.synchronize()
print(f"after a second communicator: {tiny():.1f} us; os threads: {len(os.listdir(f'/proc/{os.getpid()}/task'))}")
for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "TORCH_NCCL_BLOCKING_WAIT", "GPU_MAX_HW_QUEUES"):
    print(k, os.environ.get(k))
