"""debug: LDS canary workgroups co-resident with the conv kernels"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "debug", "libcanary.so"))
dev = torch.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
m = bench.build_model(dev, prec)
ex = m._executor(dev)
net = m.net
side = torch.cuda.Stream()
for lname, shape in (("d0c1", (2, 270, 480, 128)), ("d0c1", (1, 64, 96, 128)), ("inc3", (2, 540, 960, 64)), ("down0", (2, 540, 960, 64)), ("out3", (2, 540, 960, 64))):
    sp = net.temp1[lname] if lname != "out3" else net.temp2["out0"]
    big = torch.rand(shape, device=dev)
    blocks, lds_bytes = 512, 40 * 1024
    bad = torch.zeros(blocks, dtype=torch.int32, device=dev)
    first = torch.full((blocks,), 0x7fffffff, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = lib.canary_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(bad.data_ptr()),
                           ctypes.c_void_p(first.data_ptr()), blocks, lds_bytes, 400)
    assert rc == 0, rc
    with torch.cuda.stream(side):
        for _ in range(40):
            ex.conv(sp, big)
    torch.cuda.synchronize()
    nb = int((bad > 0).sum())
    print(prec, lname, shape, "canary workgroups with corrupted LDS: %d of %d, corrupted words %d, first offsets %s" % (
        nb, blocks, int(bad.sum()), sorted(set(first[bad > 0].tolist()))[:10]))
