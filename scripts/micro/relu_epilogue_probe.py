import torch, time
torch.manual_seed(0)
x=torch.randn(7600,256,device="cuda"); W=torch.randn(256,256,device="cuda")*0.05; b=torch.randn(256,device="cuda")
x2=torch.randn(7600,512,device="cuda"); W2=torch.randn(256,512,device="cuda")*0.05
for (xx,WW) in ((x,W),(x2,W2)):
    a=torch.relu(torch.addmm(b,xx,WW.t())); c=torch._addmm_activation(b,xx,WW.t())
    print("equal:",torch.equal(a,c),float((a-c).abs().max()))
    for name,f in (("addmm+relu",lambda: torch.relu_(torch.addmm(b,xx,WW.t()))),("_addmm_activation",lambda: torch._addmm_activation(b,xx,WW.t()))):
        for _ in range(20): f()
        torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): f()
        e.record(); torch.cuda.synchronize()
        print(name, xx.shape[1], "%.1f us"%(s.elapsed_time(e)/200*1e3))
