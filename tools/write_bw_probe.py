import torch, time
n = 65536*21168
x = torch.empty(n, dtype=torch.uint8, device="cuda")
x32 = x.view(torch.int32)
def bench(f, name, bytes_):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)/20
    print("%s: %.1f us  %.2f TB/s" % (name, ms*1e3, bytes_/ms/1e9))
bench(lambda: x.zero_(), "memset u8 (hipMemsetAsync)", n)
bench(lambda: x32.fill_(0x01020304), "fill int32 (elementwise kernel)", n)
y = torch.empty_like(x32)
bench(lambda: y.copy_(x32), "copy int32 (read+write, bytes = written)", n)
for m in (16384*21168, 32768*21168):
    z = x[:m]
    bench(lambda: z.zero_(), "memset %d MB" % (m>>20), m)
