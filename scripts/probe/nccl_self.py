# Probe: does RCCL with world_size=1 support all_reduce + batched isend/irecv to self on one GPU?
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
t = torch.tensor([3.0], device="cuda"); dist.all_reduce(t); print("allreduce ok", t.item())
a = torch.arange(8, dtype=torch.float64, device="cuda"); b = a + 100
ra = torch.zeros_like(a); rb = torch.zeros_like(a)
ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.isend, b, 0), dist.P2POp(dist.irecv, ra, 0), dist.P2POp(dist.irecv, rb, 0)]
for w in dist.batch_isend_irecv(ops): w.wait()
torch.cuda.synchronize(); print("p2p self ok", ra.tolist(), rb.tolist())
dist.destroy_process_group()
