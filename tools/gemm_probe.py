import torch, time
dev='cuda:0'
shapes=[(102400,128,1152),(25600,512,2304),(409600,128,128),(25600,512,512),(102400,256,256),(409600,128,256),(25600,256,2304),(102400,128,128),(8192,8192,8192)]
for M,N,K in shapes:
    a=torch.randn(M,K,device=dev).bfloat16(); b=torch.randn(N,K,device=dev).bfloat16()
    for _ in range(5): c=a@b.t()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    ts=[]
    for r in range(10):
        e0.record()
        for _ in range(8): c=a@b.t()
        e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1)/8*1e3)
    ts.sort(); t=ts[len(ts)//2]
    print(f"M={M} N={N} K={K}: {t:.1f} us  {2*M*N*K/t/1e6:.0f} TFLOP/s  {(M*K+N*K+M*N)*2/t/1e3:.0f} GB/s",flush=True)
