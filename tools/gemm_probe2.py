import torch, time
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    return (t1-t0)/n*1e3, (t2-t0)/n*1e3
B=131072; S=64
for dt in (torch.bfloat16, torch.float32):
    for (o,i) in ((256,144),(256,256),(12,256),(1,256)):
        x=torch.randn(B,i,device="cuda",dtype=dt); dy=torch.randn(B,o,device="cuda",dtype=dt); w=torch.randn(o,i,device="cuda",dtype=dt)
        print(dt, (o,i), "bmm dW host/total ms: %.3f %.3f" % t(lambda: torch.bmm(dy.view(S,B//S,-1).transpose(1,2), x.view(S,B//S,-1)).float().sum(0)),
              " dx: %.3f %.3f" % t(lambda: dy@w), " fwd: %.3f %.3f" % t(lambda: torch.nn.functional.linear(x,w)),
              " plain dW: %.3f %.3f" % t(lambda: dy.t()@x))
