import os, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
s = torch.cuda.Stream(device=dev, priority=-1)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    x = torch.arange(8, device=dev, dtype=torch.float32)
    out = torch.empty(8, device=dev)
    dist.all_gather_into_tensor(out, x)
    dist.all_reduce(x)
    cnt = [torch.zeros(1, dtype=torch.int64, device=dev)]
    dist.all_gather(cnt, torch.tensor([3], dtype=torch.int64, device=dev))
    dist.barrier()
    torch.cuda.synchronize()
    print('nccl on a high-priority stream ok', out.tolist(), x.tolist(), int(cnt[0]))
dist.barrier(); dist.destroy_process_group()
