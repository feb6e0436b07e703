import torch, time
dev=torch.device("cuda",0)
for mb in (4,12,32,96):
    h=torch.empty(mb<<20,dtype=torch.uint8).pin_memory()
    d=torch.empty(mb<<20,dtype=torch.uint8,device=dev)
    s=torch.cuda.Stream(dev)
    for rep in range(3):
        torch.cuda.synchronize()
        t=time.perf_counter()
        with torch.cuda.stream(s):
            for _ in range(8): d.copy_(h,non_blocking=True)
        s.synchronize()
        dt=time.perf_counter()-t
    print(f"pinned H2D {mb} MB x8: {8*mb/1024/dt:.1f} GB/s")
    hp=torch.empty(mb<<20,dtype=torch.uint8)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(4): d.copy_(hp)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f"pageable H2D {mb} MB x4: {4*mb/1024/dt:.1f} GB/s")
    t=time.perf_counter()
    for _ in range(4): h.copy_(hp)
    dt=time.perf_counter()-t
    print(f"host memcpy pageable->pinned {mb} MB x4 (1 thread): {4*mb/1024/dt:.1f} GB/s")
