import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
B=131072
for dt in (torch.float32, torch.bfloat16):
    x=torch.randn(B,144,device="cuda",dtype=dt); dy=torch.randn(B,256,device="cuda",dtype=dt); w=torch.randn(256,144,device="cuda",dtype=dt)
    print(dt, "plain dW dy.t()@x        %.1f us" % t(lambda: dy.t()@x))
    for S in (8,32,64,128):
        print(dt, f"bmm S={S} transpose(1,2)   %.1f us" % t(lambda: torch.bmm(dy.view(S,B//S,-1).transpose(1,2), x.view(S,B//S,-1)).float().sum(0)))
        print(dt, f"bmm S={S} x^T dy (then .T) %.1f us" % t(lambda: torch.bmm(x.view(S,B//S,-1).transpose(1,2), dy.view(S,B//S,-1)).float().sum(0)))
    print(dt, "fwd x@w.t()              %.1f us" % t(lambda: x@w.t()))
    print(dt, "dx dy@w                  %.1f us" % t(lambda: dy@w))
