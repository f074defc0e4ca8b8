import time, torch
x = torch.randn(3072, 3072, device='cuda')
def work():
    y = x
    for _ in range(3):
        y = y @ x
    return y
for _ in range(3): work()
torch.cuda.synchronize()
def t(name, waiter):
    ts = []
    for _ in range(8):
        work()
        ev = torch.cuda.Event(); ev.record()
        t0 = time.perf_counter(); waiter(ev); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, ' '.join('%.2f' % v for v in ts))
t('event.synchronize', lambda ev: ev.synchronize())
def spin(ev):
    while not ev.query():
        pass
t('event.query spin', spin)
t('cuda.synchronize', lambda ev: torch.cuda.synchronize())
t('stream.synchronize', lambda ev: torch.cuda.current_stream().synchronize())
def blocking():
    ts=[]
    for _ in range(8):
        work()
        ev = torch.cuda.Event(blocking=True); ev.record()
        t0 = time.perf_counter(); ev.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print('blocking event', ' '.join('%.2f' % v for v in ts))
blocking()
host = torch.empty(4, pin_memory=True); d = torch.zeros(4, device='cuda')
ts=[]
for _ in range(8):
    work(); host.copy_(d, non_blocking=True); ev = torch.cuda.Event(); ev.record()
    t0 = time.perf_counter(); ev.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('copy+event.synchronize', ' '.join('%.2f' % v for v in ts))
ts=[]
for _ in range(8):
    work()
    t0 = time.perf_counter(); d.cpu(); ts.append((time.perf_counter() - t0) * 1e3)
print('.cpu()', ' '.join('%.2f' % v for v in ts))
