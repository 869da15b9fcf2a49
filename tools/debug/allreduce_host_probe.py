import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
flat = torch.zeros(11_000_000, device=dev)
busy = torch.randn(8192, 8192, device=dev)
s = torch.cuda.Stream()
dist.all_reduce(flat); torch.cuda.synchronize()
for mode in ("idle", "busy other stream", "busy same stream"):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        if mode == "busy other stream":
            with torch.cuda.stream(s):
                for _ in range(20): busy @ busy
        elif mode == "busy same stream":
            for _ in range(20): busy @ busy
        t0 = time.perf_counter()
        dist.all_reduce(flat)
        t1 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); dist.all_reduce(flat); e1.record(); torch.cuda.synchronize()
        ts.append(((t1 - t0) * 1e3, e0.elapsed_time(e1)))
    print(mode, "host ms of the call: %.3f, device ms of an isolated call: %.3f" % (sorted(t[0] for t in ts)[2], sorted(t[1] for t in ts)[2]))
dist.destroy_process_group()
